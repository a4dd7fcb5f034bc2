"""Differentiable (torch-CPU float64) restatement of the deformable operators' FORWARD (oracle/deform.py, i.e.
nn/deformable_im2col.cuh:76-113,215-262 and deformable_psroi_pooling.cu:29-138) -- TEST INFRASTRUCTURE ONLY: autograd
of these functions is the checker of the HIP backward kernels.  Index decisions (floor / clamp / inside tests) are
computed on detached values exactly like the numpy oracle; interpolation weights stay differentiable.
tests/test_oracle_deform.py pins the forward values to oracle/deform.py.
"""
import numpy as np
import torch


def deformable_im2col(data, offset, kernel, pad, stride, dilate, dg):
    """data [C,H,W], offset [2*kh*kw*dg,Ho,Wo] torch float64 -> col [C*kh*kw, Ho, Wo] (row c*kh*kw + tap)."""
    C, H, W = data.shape
    kh, kw = kernel
    Ho = (H + 2 * pad[0] - (dilate[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dilate[1] * (kw - 1) + 1)) // stride[1] + 1
    cpg = C // dg
    h_in = (torch.arange(Ho) * stride[0] - pad[0]).view(Ho, 1).double()
    w_in = (torch.arange(Wo) * stride[1] - pad[1]).view(1, Wo).double()
    cols = []
    for c0 in range(dg):
        d = data[c0 * cpg:(c0 + 1) * cpg]
        per_tap = []
        for i in range(kh):
            for j in range(kw):
                t = i * kw + j
                oh, ow = offset[c0 * 2 * kh * kw + 2 * t], offset[c0 * 2 * kh * kw + 2 * t + 1]
                h_im = h_in + i * dilate[0] + oh
                w_im = w_in + j * dilate[1] + ow
                inside = ((h_im >= 0) & (w_im >= 0) & (h_im < H) & (w_im < W)).detach()
                h_low = torch.floor(h_im.detach()); w_low = torch.floor(w_im.detach())
                ch, cw = h_low >= H - 1, w_low >= W - 1
                h_low = torch.where(ch, torch.full_like(h_low, H - 1), h_low)
                w_low = torch.where(cw, torch.full_like(w_low, W - 1), w_low)
                h_high = torch.where(ch, h_low, h_low + 1); w_high = torch.where(cw, w_low, w_low + 1)
                lh = torch.where(ch, torch.zeros_like(h_im), h_im - h_low)
                lw = torch.where(cw, torch.zeros_like(w_im), w_im - w_low)
                ya, yb = h_low.clamp(0, H - 1).long(), h_high.clamp(0, H - 1).long()
                xa, xb = w_low.clamp(0, W - 1).long(), w_high.clamp(0, W - 1).long()
                v = ((1 - lh) * (1 - lw)) * d[:, ya, xa] + ((1 - lh) * lw) * d[:, ya, xb] + (lh * (1 - lw)) * d[:, yb, xa] + (lh * lw) * d[:, yb, xb]
                per_tap.append(torch.where(inside, v, torch.zeros_like(v)))
        cols.append(torch.stack(per_tap, 1))                      # [cpg, kh*kw, Ho, Wo]
    return torch.cat(cols, 0).reshape(C * kh * kw, Ho, Wo)


def deformable_convolution(data, offset, weight, kernel, stride, dilate, pad, dg):
    """data [N,C,H,W], offset [N,..], weight [Co,C,kh,kw] -> [N,Co,Ho,Wo]."""
    outs = []
    for n in range(data.shape[0]):
        col = deformable_im2col(data[n], offset[n], kernel, pad, stride, dilate, dg)
        K, Ho, Wo = col.shape
        outs.append((weight.reshape(weight.shape[0], K) @ col.reshape(K, -1)).reshape(-1, Ho, Wo))
    return torch.stack(outs)


def deformable_psroi_pooling(data, rois, trans, spatial_scale, output_dim, group_size, pooled_size, part_size,
                             sample_per_part, trans_std, no_trans):
    """data [B,C,H,W] float64 torch, rois numpy [R,5], trans [R,2*ncls,part,part] torch | None -> [R,output_dim,P,P]."""
    B, Cc, H, W = data.shape
    rois = np.asarray(rois, np.float32)
    R, P = rois.shape[0], pooled_size
    part = part_size or P
    F = np.float32
    ncls = 1 if no_trans else trans.shape[1] // 2
    ch_each = output_dim if no_trans else output_dim // ncls
    ph = np.arange(P)
    part_idx = np.floor((ph.astype(F) / F(P)).astype(F) * F(part)).astype(np.int64)
    g_idx = np.clip(np.floor((ph.astype(F) * F(group_size)).astype(F) / F(P)).astype(np.int64), 0, group_size - 1)
    ctop = np.arange(output_dim)
    cls = torch.as_tensor(ctop // ch_each)
    rnd = lambda x: np.sign(x) * np.floor(np.abs(x) + F(0.5))
    outs = []
    for n in range(R):
        b = int(rois[n, 0])
        rs_w = F(F(rnd(rois[n, 1]) * F(spatial_scale)) - F(0.5)); rs_h = F(F(rnd(rois[n, 2]) * F(spatial_scale)) - F(0.5))
        re_w = F(F(F(rnd(rois[n, 3]) + 1) * F(spatial_scale)) - F(0.5)); re_h = F(F(F(rnd(rois[n, 4]) + 1) * F(spatial_scale)) - F(0.5))
        rw, rh = float(max(F(re_w - rs_w), F(0.1))), float(max(F(re_h - rs_h), F(0.1)))
        bw, bh = rw / P, rh / P
        sw, sh = bw / sample_per_part, bh / sample_per_part
        if no_trans:
            tx = torch.zeros(output_dim, P, P, dtype=torch.float64); ty = torch.zeros_like(tx)
        else:
            t = trans[n].reshape(ncls, 2, part, part)
            pi = torch.as_tensor(part_idx)
            tx = t[cls, 0][:, pi][:, :, pi] * trans_std
            ty = t[cls, 1][:, pi][:, :, pi] * trans_std
        wstart = torch.as_tensor(ph * bw + float(rs_w)).view(1, 1, P) + tx * rw
        hstart = torch.as_tensor(ph * bh + float(rs_h)).view(1, P, 1) + ty * rh
        c = torch.as_tensor((ctop[:, None, None] * group_size + g_idx[None, :, None]) * group_size + g_idx[None, None, :])
        d = data[b]
        s = torch.zeros(output_dim, P, P, dtype=torch.float64)
        k = torch.zeros(output_dim, P, P, dtype=torch.float64)
        for ih in range(sample_per_part):
            for iw in range(sample_per_part):
                w = wstart + iw * sw
                h = hstart + ih * sh
                ok = ~((w < -0.5) | (w > W - 0.5) | (h < -0.5) | (h > H - 0.5)).detach()
                w = w.clamp(0, W - 1); h = h.clamp(0, H - 1)
                x1 = torch.floor(w.detach()).long(); x2 = torch.ceil(w.detach()).long()
                y1 = torch.floor(h.detach()).long(); y2 = torch.ceil(h.detach()).long()
                dx, dy = w - x1, h - y1
                val = (1 - dx) * (1 - dy) * d[c, y1, x1] + (1 - dx) * dy * d[c, y2, x1] + dx * (1 - dy) * d[c, y1, x2] + dx * dy * d[c, y2, x2]
                s = s + torch.where(ok, val, torch.zeros_like(val))
                k = k + ok.double()
        outs.append(torch.where(k > 0, s / k.clamp(min=1), torch.zeros_like(s)))
    return torch.stack(outs)
