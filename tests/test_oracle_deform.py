"""CPU checks of oracle/deform.py (the deformable operators' restatement; parity unpinned, see its header).

The vectorised oracle is compared, bit for bit, with scalar twins written here statement by statement
after the reference's CUDA kernels (deformable_im2col.cuh:76-113,215-262; deformable_psroi_pooling.cu:29-138),
and with torch's ordinary convolution where the deformable operator must degenerate to it."""
import math
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import deform  # noqa: E402

F = np.float32


def scalar_im2col(data, offset, kernel, pad, stride, dilate, dg):
    C, H, W = data.shape
    kh, kw = kernel
    Ho = (H + 2 * pad[0] - (dilate[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dilate[1] * (kw - 1) + 1)) // stride[1] + 1
    col = np.zeros((C * kh * kw, Ho, Wo), F)
    cpg = C // dg
    for c_im in range(C):
        g = c_im // cpg
        for h_col in range(Ho):
            for w_col in range(Wo):
                h_in = h_col * stride[0] - pad[0]
                w_in = w_col * stride[1] - pad[1]
                for i in range(kh):
                    for j in range(kw):
                        off_h = offset[g * 2 * kh * kw + 2 * (i * kw + j), h_col, w_col]
                        off_w = offset[g * 2 * kh * kw + 2 * (i * kw + j) + 1, h_col, w_col]
                        val = F(0)
                        h_im = F(F(h_in + i * dilate[0]) + off_h)
                        w_im = F(F(w_in + j * dilate[1]) + off_w)
                        if h_im >= 0 and w_im >= 0 and h_im < H and w_im < W:
                            h = F(F(i * dilate[0]) + off_h)
                            w = F(F(j * dilate[1]) + off_w)
                            height, width = H - h_in, W - w_in
                            h_low, w_low = int(math.floor(h)), int(math.floor(w))
                            if h_low >= height - 1:
                                h_high = h_low = height - 1
                                h = F(h_low)
                            else:
                                h_high = h_low + 1
                            if w_low >= width - 1:
                                w_high = w_low = width - 1
                                w = F(w_low)
                            else:
                                w_high = w_low + 1
                            lh, lw = F(h - F(h_low)), F(w - F(w_low))
                            hh, hw = F(F(1) - lh), F(F(1) - lw)
                            v1 = data[c_im, h_in + h_low, w_in + w_low]
                            v2 = data[c_im, h_in + h_low, w_in + w_high]
                            v3 = data[c_im, h_in + h_high, w_in + w_low]
                            v4 = data[c_im, h_in + h_high, w_in + w_high]
                            w1, w2, w3, w4 = F(hh * hw), F(hh * lw), F(lh * hw), F(lh * lw)
                            val = F(F(F(F(w1 * v1) + F(w2 * v2)) + F(w3 * v3)) + F(w4 * v4))
                        col[c_im * kh * kw + i * kw + j, h_col, w_col] = val
    return col


def scalar_psroi(data, rois, trans, scale, output_dim, group, P, part, spp, trans_std, no_trans):
    B, C, H, W = data.shape
    R = rois.shape[0]
    scale, trans_std = F(scale), F(trans_std)
    num_classes = 1 if no_trans else trans.shape[1] // 2
    ch_each = output_dim if no_trans else output_dim // num_classes
    out = np.zeros((R, output_dim, P, P), F)
    cnt = np.zeros((R, output_dim, P, P), F)

    def rnd(x):
        return F(math.floor(abs(float(x)) + 0.5) * (1 if x >= 0 else -1))

    for n in range(R):
        for ctop in range(output_dim):
            for ph in range(P):
                for pw in range(P):
                    b = int(rois[n, 0])
                    rsw = F(F(rnd(rois[n, 1]) * scale) - F(0.5)); rsh = F(F(rnd(rois[n, 2]) * scale) - F(0.5))
                    rew = F(F(F(rnd(rois[n, 3]) + F(1)) * scale) - F(0.5)); reh = F(F(F(rnd(rois[n, 4]) + F(1)) * scale) - F(0.5))
                    rw = max(F(rew - rsw), F(0.1)); rh = max(F(reh - rsh), F(0.1))
                    bh, bw = F(rh / F(P)), F(rw / F(P))
                    sh, sw = F(bh / F(spp)), F(bw / F(spp))
                    part_h = int(math.floor(F(F(F(ph) / F(P)) * F(part))))
                    part_w = int(math.floor(F(F(F(pw) / F(P)) * F(part))))
                    cls = ctop // ch_each
                    tx = F(0) if no_trans else F(trans[n, cls * 2, part_h, part_w] * trans_std)
                    ty = F(0) if no_trans else F(trans[n, cls * 2 + 1, part_h, part_w] * trans_std)
                    wstart = F(F(F(pw) * bw) + rsw); wstart = F(wstart + F(tx * rw))
                    hstart = F(F(F(ph) * bh) + rsh); hstart = F(hstart + F(ty * rh))
                    s, k = F(0), 0
                    gw = min(max(int(math.floor(F(F(F(pw) * F(group)) / F(P)))), 0), group - 1)
                    gh = min(max(int(math.floor(F(F(F(ph) * F(group)) / F(P)))), 0), group - 1)
                    for ih in range(spp):
                        for iw in range(spp):
                            w = F(wstart + F(F(iw) * sw)); h = F(hstart + F(F(ih) * sh))
                            if w < -0.5 or w > W - 0.5 or h < -0.5 or h > H - 0.5:
                                continue
                            w = F(min(max(w, F(0)), F(W - 1))); h = F(min(max(h, F(0)), F(H - 1)))
                            c = (ctop * group + gh) * group + gw
                            x1, x2, y1, y2 = int(math.floor(w)), int(math.ceil(w)), int(math.floor(h)), int(math.ceil(h))
                            dx, dy = F(w - F(x1)), F(h - F(y1))
                            d = data[b, c]
                            one = F(1)
                            val = F(F(F(F(one - dx) * F(one - dy)) * d[y1, x1]) + F(F(F(one - dx) * dy) * d[y2, x1]))
                            val = F(val + F(F(dx * F(one - dy)) * d[y1, x2]))
                            val = F(val + F(F(dx * dy) * d[y2, x2]))
                            s = F(s + val); k += 1
                    out[n, ctop, ph, pw] = F(0) if k == 0 else F(s / F(k))
                    cnt[n, ctop, ph, pw] = k
    return out, cnt


def test_im2col_matches_scalar_twin():
    rng = np.random.default_rng(0)
    for (C, H, W, k, pad, st, dil, dg, sig) in [(4, 7, 9, 3, 2, 1, 2, 2, 1.5), (2, 6, 5, 3, 1, 2, 1, 1, 3.0)]:
        data = rng.normal(0, 1, (C, H, W)).astype(F)
        Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // st + 1
        Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // st + 1
        off = rng.normal(0, sig, (2 * k * k * dg, Ho, Wo)).astype(F)
        a = deform.deformable_im2col(data, off, (k, k), (pad, pad), (st, st), (dil, dil), dg)
        b = scalar_im2col(data, off, (k, k), (pad, pad), (st, st), (dil, dil), dg)
        assert a.shape == b.shape and np.array_equal(a, b)
        assert (a == 0).mean() > 0.02           # some samples fall outside the image


def test_zero_and_integer_offsets_reduce_to_convolution():
    rng = np.random.default_rng(1)
    N, C, H, W, Co, k, pad, dil = 2, 8, 10, 12, 6, 3, 2, 2
    data = rng.normal(0, 1, (N, C, H, W)).astype(F)
    wgt = rng.normal(0, 0.2, (Co, C, k, k)).astype(F)
    bias = rng.normal(0, 0.1, Co).astype(F)
    off = np.zeros((N, 2 * k * k * 4, H, W), F)
    ref = torch.nn.functional.conv2d(torch.as_tensor(data), torch.as_tensor(wgt), torch.as_tensor(bias), padding=pad, dilation=dil).numpy()
    out = deform.deformable_convolution(data, off, wgt, bias, (k, k), (1, 1), (dil, dil), (pad, pad), num_deformable_group=4)
    assert np.abs(out - ref).max() <= 1e-5
    # a constant integer offset (+1 row, -2 columns) = the plain convolution read at shifted positions of the
    # zero-extended image: convolve a canvas with margin m and slice the shifted window
    off[:, 0::2] = 1.0
    off[:, 1::2] = -2.0
    m = 4
    canvas = np.zeros((N, C, H + 2 * m, W + 2 * m), F)
    canvas[:, :, m:m + H, m:m + W] = data
    full = torch.nn.functional.conv2d(torch.as_tensor(canvas), torch.as_tensor(wgt), None, padding=0, dilation=dil).numpy()
    ref = full[:, :, m - pad + 1:m - pad + 1 + H, m - pad - 2:m - pad - 2 + W]
    out = deform.deformable_convolution(data, off, wgt, None, (k, k), (1, 1), (dil, dil), (pad, pad), num_deformable_group=4)
    assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-5


def test_psroi_matches_scalar_twin():
    rng = np.random.default_rng(2)
    B, od, H, W, P = 2, 4, 9, 11, 3
    rois = np.array([[0, 10.3, 20.7, 90.2, 100.4], [1, -30, -10, 40, 60.5], [0, 100, 80, 400, 300], [1, 50, 50, 50, 50]], F)
    for group, no_trans, ncls in [(1, True, 1), (1, False, 1), (2, False, 2)]:
        data = rng.normal(0, 1, (B, od * group * group, H, W)).astype(F)
        trans = None if no_trans else rng.normal(0, 1.5, (4, 2 * ncls, P, P)).astype(F)
        a, ac = deform.deformable_psroi_pooling(data, rois, trans, 0.0625, od, group, P, P, 2, 0.1, no_trans)
        b, bc = scalar_psroi(data, rois, trans, 0.0625, od, group, P, P, 2, 0.1, no_trans)
        assert np.array_equal(ac, bc)
        assert np.array_equal(a, b)
    assert (ac == 0).any() or (ac < 4).any()     # the out-of-image roi drops samples


def test_psroi_part_index_uses_fp32_division():
    # floor(float(ph) / 7 * 7) is not ph for every ph in fp32 (deformable_psroi_pooling.cu:92): the oracle
    # must reproduce the reference's part index, not the "obvious" one.
    ph = np.arange(7)
    idx = np.floor((ph.astype(F) / F(7)).astype(F) * F(7)).astype(int)
    assert idx.tolist() == [int(math.floor(F(F(F(p) / F(7)) * F(7)))) for p in range(7)]


def test_torch_restatements_match_numpy_oracle():
    """oracle/deform_torch.py (autograd checker of the backward kernels) == oracle/deform.py on the forward."""
    from oracle import deform_torch as DT
    rng = np.random.default_rng(5)
    C, H, W, k, pad, dil, dg = 8, 9, 11, 3, 2, 2, 2
    data = rng.normal(0, 1, (C, H, W)).astype(F)
    off = rng.normal(0, 2.0, (2 * k * k * dg, H, W)).astype(F)
    a = deform.deformable_im2col(data, off, (k, k), (pad, pad), (1, 1), (dil, dil), dg)
    b = DT.deformable_im2col(torch.as_tensor(data).double(), torch.as_tensor(off).double(), (k, k), (pad, pad), (1, 1), (dil, dil), dg).numpy()
    assert np.abs(a - b).max() <= 1e-5
    rois = np.array([[0, 10.3, 20.7, 90.2, 100.4], [1, -30, -10, 40, 60.5], [0, 100, 80, 400, 300]], F)
    d4 = rng.normal(0, 1, (2, 8, 9, 11)).astype(F)
    trans = rng.normal(0, 1.5, (3, 4, 3, 3)).astype(F)
    a, _ = deform.deformable_psroi_pooling(d4, rois, trans, 0.0625, 2, 2, 3, 3, 2, 0.1, False)
    b = DT.deformable_psroi_pooling(torch.as_tensor(d4).double(), rois, torch.as_tensor(trans).double(), 0.0625, 2, 2, 3, 3, 2, 0.1, False).numpy()
    assert np.abs(a - b).max() <= 1e-5
