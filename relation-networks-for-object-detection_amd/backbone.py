"""ResNet-101 conv1..conv5 + RPN head + conv_new_1 of the reference graph
(relation_rcnn/symbols/resnet_v1_101_rcnn_base.py:29-693, SYM_REL:249-250).

impl='hip' (bf16, the throughput path): every convolution, the 7x7 stem included, is a kernel of librelnet_hip.so over
NHWC activations -- the implicit-GEMM MFMA kernels of csrc/gemm.hip with bias / ReLU / shortcut fused into the epilogue,
the halo-resident 3x3 kernel and the block-boundary chain kernel of csrc/bottleneck.hip for res2; no library call.
impl='hip32' (float32 parity path, the default for dtype float32): the same graph on the exact-fp32 MFMA convolution kernel
(relnet_conv2d_nhwc_f32: an fmaf chain per output element) + relnet_maxpool_nhwc_f32 -- no library call either.
impl='miopen' runs the graph through torch.nn.functional.conv2d (kept for the A/B against the library).  What this module owns is the
graph itself: the Caffe-style ResNet (stride on the FIRST 1x1 of a stage, :99,103), conv5 dilated 2 with stride 1
(:632-633), ceil-mode pool1 (pooling_convention='full', :35-36) and the frozen BatchNorm (use_global_stats=True, eps=1e-5,
:32) folded into the preceding convolution at load time.
Parameter names are the reference's (`res4b7_branch2a_weight`, `bn4b7_branch2a_gamma`, ...).
"""
import math
import os

import torch
import torch.nn.functional as F

from . import ops

UNITS = (3, 4, 23, 3)
FILTERS = (256, 512, 1024, 2048)
EPS = 1e-5


def unit_names(fpn=False):
    """[(stage, unit_name, in_ch, mid_ch, out_ch, stride, dilate, has_proj)] in graph order.  fpn: res5 has
    stride 2 and no dilation (symbols/resnet_v1_101_rcnn_fpn_..._learn_nms.py:707-720)."""
    out = []
    in_ch = 64
    for si, (n, oc) in enumerate(zip(UNITS, FILTERS)):
        stage = si + 2
        for u in range(n):
            if n <= 3 or stage == 2 or stage == 5:
                name = 'abc'[u]
            else:
                name = 'a' if u == 0 else 'b%d' % u
            stride = 2 if (u == 0 and (stage in (3, 4) or (fpn and stage == 5))) else 1
            dilate = 2 if (stage == 5 and not fpn) else 1
            out.append((stage, '%d%s' % (stage, name), in_ch, oc // 4, oc, stride, dilate, u == 0))
            in_ch = oc
    return out


def conv_bn_names():
    """[(conv_name, bn_name, out_ch, in_ch, k)] for every conv+BN pair of the backbone."""
    layers = [('conv1', 'bn_conv1', 64, 3, 7)]
    for stage, nm, ic, mc, oc, stride, dil, proj in unit_names():
        if proj:
            layers.append(('res%s_branch1' % nm, 'bn%s_branch1' % nm, oc, ic, 1))
        layers.append(('res%s_branch2a' % nm, 'bn%s_branch2a' % nm, mc, ic, 1))
        layers.append(('res%s_branch2b' % nm, 'bn%s_branch2b' % nm, mc, mc, 3))
        layers.append(('res%s_branch2c' % nm, 'bn%s_branch2c' % nm, oc, mc, 1))
    return layers


def init_params(seed=1, num_anchors=12, num_classes=81, num_reg_classes=2, generator=None, dcn_offset_std=0.0, fpn=False):
    """Random-init parameters of the relation test graph under the reference's names.
    New layers N(0, 0.01)/0 as init_weight (SYM_REL:327-362); the backbone has no pretrained
    weights offline: He-normal convs, BN gamma=1 (0.25 on the residual branch output so the
    34 residual adds keep activations O(1)), beta=0, mean=0, var=1."""
    g = generator or torch.Generator().manual_seed(seed)
    p = {}

    def nrm(std, *shape):
        return torch.randn(*shape, generator=g) * std

    for conv, bn, oc, ic, k in conv_bn_names():
        p[conv + '_weight'] = nrm(math.sqrt(2.0 / (ic * k * k)), oc, ic, k, k)
        p[bn + '_gamma'] = torch.full((oc,), 0.25 if conv.endswith('branch2c') else 1.0)
        p[bn + '_beta'] = torch.zeros(oc)
        p[bn + '_moving_mean'] = torch.zeros(oc)
        p[bn + '_moving_var'] = torch.ones(oc)
    p['rpn_conv_3x3_weight'] = nrm(0.01, 512, 1024, 3, 3); p['rpn_conv_3x3_bias'] = torch.zeros(512)
    p['rpn_cls_score_weight'] = nrm(0.01, 2 * num_anchors, 512, 1, 1); p['rpn_cls_score_bias'] = torch.zeros(2 * num_anchors)
    p['rpn_bbox_pred_weight'] = nrm(0.01, 4 * num_anchors, 512, 1, 1); p['rpn_bbox_pred_bias'] = torch.zeros(4 * num_anchors)
    p['conv_new_1_weight'] = nrm(0.01, 256, 2048, 1, 1); p['conv_new_1_bias'] = torch.zeros(256)
    p['fc_new_1_weight'] = nrm(0.01, 1024, 256 * 49); p['fc_new_1_bias'] = torch.zeros(1024)
    p['fc_new_2_weight'] = nrm(0.01, 1024, 1024); p['fc_new_2_bias'] = torch.zeros(1024)
    p['cls_score_weight'] = nrm(0.01, num_classes, 1024); p['cls_score_bias'] = torch.zeros(num_classes)
    p['bbox_pred_weight'] = nrm(0.01, 4 * num_reg_classes, 1024); p['bbox_pred_bias'] = torch.zeros(4 * num_reg_classes)
    for i in (1, 2):
        p['pair_pos_fc1_%d_weight' % i] = nrm(0.01, 16, 64); p['pair_pos_fc1_%d_bias' % i] = torch.zeros(16)
        p['query_%d_weight' % i] = nrm(0.01, 1024, 1024); p['query_%d_bias' % i] = torch.zeros(1024)
        p['key_%d_weight' % i] = nrm(0.01, 1024, 1024); p['key_%d_bias' % i] = torch.zeros(1024)
        p['linear_out_%d_weight' % i] = nrm(0.01, 1024, 1024, 1, 1); p['linear_out_%d_bias' % i] = torch.zeros(1024)
    # learn-NMS head, init_weight_nms (symbols/..._learn_nms.py:571-600): N(0,0.01) / 0, logit bias -3
    p['nms_rank_weight'] = nrm(0.01, 128, 1024); p['nms_rank_bias'] = torch.zeros(128)
    p['roi_feat_embedding_weight'] = nrm(0.01, 128, 1024); p['roi_feat_embedding_bias'] = torch.zeros(128)
    p['nms_pair_pos_fc1_1_weight'] = nrm(0.01, 16, 64); p['nms_pair_pos_fc1_1_bias'] = torch.zeros(16)
    p['nms_query_1_weight'] = nrm(0.01, 1024, 128); p['nms_query_1_bias'] = torch.zeros(1024)
    p['nms_key_1_weight'] = nrm(0.01, 1024, 128); p['nms_key_1_bias'] = torch.zeros(1024)
    p['nms_linear_out_1_weight'] = nrm(0.01, 128, 128, 1, 1); p['nms_linear_out_1_bias'] = torch.zeros(128)
    p['nms_logit_weight'] = nrm(0.01, 5, 128); p['nms_logit_bias'] = torch.full((5,), -3.0)
    # DCN configuration (symbols/resnet_v1_101_rcnn_dcn_..._learn_nms.py:700-746,1075).  The reference
    # initialises every offset predictor to ZERO (:1525-1535); `dcn_offset_std` > 0 gives non-trivial offsets
    # for tests / benches that have no trained weights.
    for u in 'abc':
        p['res5%s_branch2b_offset_weight' % u] = nrm(dcn_offset_std, 72, 512, 3, 3)
        p['res5%s_branch2b_offset_bias' % u] = torch.zeros(72)
    p['offset_weight'] = nrm(dcn_offset_std, 98, 256 * 49); p['offset_bias'] = torch.zeros(98)
    if fpn:     # symbols/...fpn...:1450-1470, 1508-1511: N(0, 0.01) / 0
        for lvl, cin in ((32, 2048), (16, 1024), (8, 512), (4, 256)):
            p['fpn_ft%d_1x1_weight' % lvl] = nrm(0.01, 256, cin, 1, 1); p['fpn_ft%d_1x1_bias' % lvl] = torch.zeros(256)
            p['fpn_ft%d_3x3_weight' % lvl] = nrm(0.01, 256, 256, 3, 3); p['fpn_ft%d_3x3_bias' % lvl] = torch.zeros(256)
        p['fpn_ft64_3x3_weight'] = nrm(0.01, 256, 256, 3, 3); p['fpn_ft64_3x3_bias'] = torch.zeros(256)
        p['roi_pool_fc1_weight'] = nrm(0.01, 1024, 256 * 49); p['roi_pool_fc1_bias'] = torch.zeros(1024)
        p['roi_pool_fc2_weight'] = nrm(0.01, 1024, 1024); p['roi_pool_fc2_bias'] = torch.zeros(1024)
    return p


def fold_bn(w, gamma, beta, mean, var, eps=EPS):
    """BatchNorm(use_global_stats=True, fix_gamma=False): y = (x-mean)/sqrt(var+eps)*gamma+beta."""
    s = gamma.double() / torch.sqrt(var.double() + eps)
    return (w.double() * s.view(-1, 1, 1, 1)).float(), (beta.double() - mean.double() * s).float()


class Backbone(object):
    """Folded, device-resident backbone + RPN head + conv_new_1.

    impl='hip' (bf16 only): every convolution except the 7x7/Cin=3 stem runs on the implicit-GEMM
    MFMA kernel of csrc/gemm.hip over NHWC activations, with bias + ReLU and the bottleneck's
    residual add + ReLU fused into the GEMM epilogue (the library path spends as long in separate
    bias / add / clamp passes as in the convolutions themselves).  impl='miopen' keeps every conv
    in torch.nn.functional.conv2d (used for the float32 parity path)."""

    def __init__(self, params, dtype=torch.bfloat16, device='cuda', channels_last=True, impl=None, stem='hip', dcn=False,
                 fpn=False, chain=True, frozen_only=False):
        """frozen_only: pack conv1 + res2 only (what `forward_res2` runs): the part the training step never updates
        (cfgs/*.yaml FIXED_PARAMS) -- the Trainer keeps res3 .. heads in its own flat buffers, a second bf16 copy of them here would
        be a few hundred MB of stale weights."""
        self.dtype, self.device, self.dcn, self.fpn = dtype, device, dcn, fpn
        self.frozen_only = frozen_only
        self.use_chain = chain
        env = os.environ.get('RELNET_STAGE_SPLIT')        # A/B knob: '4:2,5:4' = stage:sub-batches ('0' / unset: no split)
        if env not in (None, '', '0'):
            self.stage_split = {int(a): int(b) for a, b in (kv.split(':') for kv in env.split(','))}
        if os.environ.get('RELNET_INPLACE_EXPAND') is not None:
            self.inplace_expand = os.environ['RELNET_INPLACE_EXPAND'] not in ('', '0')
        self.impl = impl or ('hip' if dtype == torch.bfloat16 else ('hip32' if dtype == torch.float32 and torch.device(device).type == 'cuda' else 'miopen'))
        if self.impl == 'hip' and torch.device(device).type == 'cuda':
            ops.asm_selfcheck()          # tile 19 (AGPR accumulators across asm statements) against the compiler-scheduled tile, once per process
        self.stem = stem
        assert self.impl in ('hip', 'hip32', 'miopen') and (self.impl != 'hip' or dtype == torch.bfloat16) and (self.impl != 'hip32' or dtype == torch.float32)
        self.mf = torch.channels_last if channels_last else torch.contiguous_format
        self.w = {}
        self.wp = {}
        self.wf = {}          # name -> fragment-order weight copy (panel kernel)
        self.b32 = {}
        self.wp32 = {}        # impl 'hip32': name -> ([Cout, k*k*Cin] fp32 packed weight, fp32 bias, k)
        for conv, bn, oc, ic, k in conv_bn_names():
            if frozen_only and not conv.startswith(('conv1', 'res2')):
                continue
            w, b = fold_bn(params[conv + '_weight'], params[bn + '_gamma'], params[bn + '_beta'],
                           params[bn + '_moving_mean'], params[bn + '_moving_var'])
            self._put(conv, w, b)
        if frozen_only:
            pass
        elif fpn:     # FPN neck instead of the RPN head / conv_new_1 (HAS_RPN: false in the FPN relation configs)
            for lvl in (32, 16, 8, 4):
                for k in ('1x1', '3x3'):
                    name = 'fpn_ft%d_%s' % (lvl, k)
                    self._put(name, params[name + '_weight'], params[name + '_bias'])
        else:
            for name in ('rpn_conv_3x3', 'rpn_cls_score', 'rpn_bbox_pred', 'conv_new_1'):
                self._put(name, params[name + '_weight'], params[name + '_bias'])
            # both 1x1 RPN outputs in one convolution: 24 score + 48 delta channels
            self._put('rpn_out', torch.cat([params['rpn_cls_score_weight'], params['rpn_bbox_pred_weight']], 0),
                      torch.cat([params['rpn_cls_score_bias'], params['rpn_bbox_pred_bias']], 0))
        self.units = [u for u in unit_names(fpn) if not frozen_only or u[0] == 2]
        self.wp_dcn = {}
        if dcn and not frozen_only:     # res5{a,b,c}_branch2b become DeformableConvolution(num_deformable_group=4) fed by a 72-channel offset conv
            for u in 'abc':
                name = 'res5%s_branch2b' % u
                self._put(name + '_offset', params[name + '_offset_weight'], params[name + '_offset_bias'])
                w, b = fold_bn(params[name + '_weight'], params['bn5%s_branch2b_gamma' % u], params['bn5%s_branch2b_beta' % u],
                               params['bn5%s_branch2b_moving_mean' % u], params['bn5%s_branch2b_moving_var' % u])
                self.wp_dcn[name] = (ops.pack_conv_weight(w, self.dtype, self.device), b.to(self.device, torch.float32).contiguous())
        if self.impl == 'hip':
            w1, b1 = fold_bn(params['conv1_weight'], params['bn_conv1_gamma'], params['bn_conv1_beta'],
                             params['bn_conv1_moving_mean'], params['bn_conv1_moving_var'])
            self.w_stem = ops.pack_stem_weight(w1, self.dtype, self.device)
            self.zero_bias64 = torch.zeros(64, device=self.device, dtype=torch.float32)
        # block boundaries inside a stage (identity shortcut, stride 1) of the HBM-bound stages: expand + shortcut + ReLU of
        # unit u and reduce + ReLU of unit u+1 as one pixel-wise kernel (ops.bottleneck_chain); unit -> its operands
        self.chain, self.halo3, self.chain_proj = {}, {}, {}
        if self.impl == 'hip' and chain:
            # (A/B knob: RELNET_CHAIN_REDUCE256=0 keeps res4's reduce layers as their own launches, the round-3 form)
            reduce_mids = tuple(m for m in ops.CHAIN_MIDS if m != 256 or os.environ.get('RELNET_CHAIN_REDUCE256', '1') != '0')
            for (st, nm, ic, mc, oc, stride, dil, proj), nxt in zip(self.units, self.units[1:] + [None]):
                if nxt is not None and nxt[0] == st and not nxt[7] and mc in reduce_mids:
                    w3, b3, _ = self.wp['res%s_branch2c' % nm]
                    w1n, b1n, _ = self.wp['res%s_branch2a' % nxt[1]]
                    self.chain[nm] = (ops.pack_w_frag(w3), ops.pack_chain_w1(w1n), b3, b1n)
                elif mc in ops.CHAIN_EXPAND_MIDS:   # last unit of a stage, and every res4 unit: the kernel without the second product
                    w3, b3, _ = self.wp['res%s_branch2c' % nm]
                    self.chain[nm] = (ops.pack_w_frag(w3), None, b3, None)
            # first unit of res2: its stride-1 projection shortcut rides in the expand product (ops.bottleneck_chain_proj)
            for (st, nm, ic, mc, oc, stride, dil, proj) in self.units:
                if proj and stride == 1 and mc == 64 and ic == 64 and nm in self.chain:
                    w3f, w1f, b3, b1n = self.chain[nm]
                    wp_, bp_, _ = self.wp['res%s_branch1' % nm]
                    self.chain_proj[nm] = (w3f, ops.pack_w_frag(wp_), w1f, (b3 + bp_).contiguous(), b1n)
            # 64-channel 3x3 convolutions (res2 branch2b): halo tile resident in LDS instead of one LDS fill per tap
            for st, nm, ic, mc, oc, stride, dil, proj in self.units:
                if mc in ops.HALO3_CHANNELS and dil == 1 and not (self.dcn and st == 5):
                    w, b, _ = self.wp['res%s_branch2b' % nm]
                    self.halo3['res%s_branch2b' % nm] = (ops.pack_w_frag(w, panel_only=False), b)

    def _put(self, name, w, b):
        self.w[name] = (w.to(self.device, self.dtype).contiguous(memory_format=self.mf),
                        b.to(self.device, self.dtype))
        if self.impl == 'hip':
            self.b32[name] = b.to(self.device, torch.float32).contiguous()
        if self.impl == 'hip32':
            if w.shape[1] % 16:             # the 3-channel stem: zero input channels up to 16 (the kernel's k granularity)
                wz = torch.zeros((w.shape[0], (w.shape[1] + 15) // 16 * 16) + tuple(w.shape[2:]), dtype=w.dtype)
                wz[:, :w.shape[1]] = w
                w = wz
            self.wp32[name] = (ops.pack_conv_weight(w, torch.float32, self.device), b.to(self.device, torch.float32).contiguous(), int(w.shape[2]))
        if self.impl == 'hip' and w.shape[1] % 64 == 0:
            self.wp[name] = (ops.pack_conv_weight(w, self.dtype, self.device),
                             b.to(self.device, torch.float32).contiguous(), int(w.shape[2]))
            # 256-deep expand convolutions (res4 branch2c, 23 per step): a second copy of the weights in MFMA-fragment
            # order lets relnet_conv2d_nhwc_wf pick the panel kernel (A resident in LDS, W streamed through registers)
            if self.dtype == torch.bfloat16 and tuple(w.shape[1:]) == (256, 1, 1) and w.shape[0] % 256 == 0 \
                    and name.endswith('_branch2c'):
                self.wf[name] = ops.pack_w_frag(self.wp[name][0])

    def _hconv(self, x, name, stride=1, pad=0, dil=1, relu=False, resid=None, out_dtype=None):
        w, b, k = self.wp[name]
        return ops.conv2d_nhwc(x, w, b, ksize=k, stride=stride, pad=pad, dil=dil, relu=relu, resid=resid,
                               out_dtype=out_dtype, w_frag=self.wf.get(name))

    def forward_res2(self, data):
        """Stem + res2 only (the part the reference freezes in training: cfgs/*.yaml FIXED_PARAMS conv1 / res2): raw NCHW image ->
        res2c output, NHWC bf16, on the inference kernels (fused stem, halo 3x3, chain kernels incl. res2a's in-kernel projection)."""
        if self.impl == 'hip32':           # float32 parity path (frozen_only backbones return the res2c map)
            assert self.frozen_only
            return self._forward_hip32(data)
        x = ops.stem_fused(data, self.w_stem, self.b32['conv1'])
        y_next = None
        self.last_chain_units = []
        for unit in self.units:
            if unit[0] != 2:
                break
            x, y_next = self._unit_hip(x, unit, y_next)
        return x

    def _forward_hip(self, data, rpn_hook=None):
        if self.stem == 'hip':
            # conv1 7x7/2 + bias + ReLU + pool1 in ONE kernel, raw NCHW image -> pooled NHWC map (no conv map in HBM)
            x = ops.stem_fused(data, self.w_stem, self.b32['conv1'])
        elif self.stem == 'hip3':
            # the three-launch form: repack to padded NHWC4, 7x7/2 conv + bias + ReLU on the MFMA kernel, then pool1
            x = ops.stem_conv7(data, self.w_stem, self.b32['conv1'], relu=True)
            x = ops.stem_bias_relu_pool(x, self.zero_bias64)
        else:
            # library 7x7 convolution (no bias), then ONE kernel for bias + ReLU + pool1
            x = data.to(self.dtype).contiguous(memory_format=self.mf)
            x = F.conv2d(x, self.w['conv1'][0], None, stride=2, padding=3)
            x = ops.stem_bias_relu_pool(x.permute(0, 2, 3, 1), self.b32['conv1'])
        conv4 = None
        ends = {}
        y_next = None
        side, hooked = None, None
        self.last_chain_units, self.last_side_stream, self.last_stage_split = [], False, {}      # what actually ran (reported by oracle/parity.py)
        ui = 0
        while ui < len(self.units):
            unit = self.units[ui]
            stage, proj = unit[0], unit[7]
            if stage == 5 and conv4 is None:
                conv4 = x
                if rpn_hook is not None and not self.fpn:      # fork: RPN head + hook next to res5
                    main = torch.cuda.current_stream()
                    side = self._side_stream()
                    self.last_side_stream = True
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        hooked = self._rpn_head(conv4)
                        hooked = hooked + (rpn_hook(hooked[0], hooked[1]),)
            if proj and stage > 2:
                ends[stage - 1] = x
            nsplit = self._stage_split(stage, x, unit) if proj else 1
            if nsplit > 1:
                # Infinity-Cache-sized sub-batches (see _stage_split): all units of the stage on images [lo, hi), then the next range
                stage_units = [u for u in self.units if u[0] == stage]
                x = self._run_stage_split(x, stage_units, nsplit)
                ui += len(stage_units)
                continue
            x, y_next = self._unit_hip(x, unit, y_next, inplace=self.inplace_expand)
            ui += 1
        conv5 = x
        nchw = lambda t: t.permute(0, 3, 1, 2)
        if self.fpn:
            # top-down pathway: 1x1 laterals, nearest 2x upsampling + sum (one in-place kernel), 3x3 output convs
            tops = {32: self._hconv(conv5, 'fpn_ft32_1x1')}
            for lvl, src in ((16, ends[4]), (8, ends[3]), (4, ends[2])):
                tops[lvl] = ops.upsample2x_add_(self._hconv(src, 'fpn_ft%d_1x1' % lvl), tops[lvl * 2])
            out = {'fpn_ft%d' % lvl: nchw(self._hconv(tops[lvl], 'fpn_ft%d_3x3' % lvl, pad=1)) for lvl in (4, 8, 16, 32)}
            out.update(conv4=nchw(conv4), conv5=nchw(conv5))
            return out
        feat = self._hconv(conv5, 'conv_new_1', relu=True)
        out = dict(conv4=nchw(conv4), conv5=nchw(conv5), conv_new_1_relu=nchw(feat))
        if hooked is not None:
            torch.cuda.current_stream().wait_stream(side)      # join
            out.update(rpn_cls_score=hooked[0], rpn_bbox_pred=hooked[1], rpn_hook=hooked[2])
        else:
            out['rpn_cls_score'], out['rpn_bbox_pred'] = self._rpn_head(conv4)
        return out

    #: stage -> sub-batches (default: none).  Measured in round 4 and NOT adopted: running the 23 res4 units on two 27-image halves so
    #: that the 1024-channel activation (265 MB at 54 images, just over the 256 MB Infinity Cache) stays cache-resident.  The
    #: memory-side cache does keep a rewritten buffer (in-place elementwise pass: 7.0-7.3 TB/s up to 256 MB against 6.0 TB/s
    #: beyond), but that is only +20 % on the HBM rate, and half-size launches lose more at their seams: 21.8 / 22.05 ms unsplit
    #: against 22.2 / 22.4 ms split on the same box (tools/llc_probe.py, tools/scripts/r04_ab_split.sh).  Kept as an A/B knob:
    #: RELNET_STAGE_SPLIT=4:2.
    stage_split = None
    #: x_next written over the shortcut operand by the expand kernels (inference only: a unit's input is destroyed once its reduce
    #: convolution has read it).  Halves the activation footprint of a stage; RELNET_INPLACE_EXPAND=0 for the A/B.
    inplace_expand = True

    def _stage_split(self, stage, x, unit):
        return int(self.stage_split.get(stage, 1)) if self.stage_split else 1

    def _run_stage_split(self, x, stage_units, nsplit):
        B = x.shape[0]
        stage = stage_units[0][0]
        st, nm, ic, mc, oc, stride, dil, proj = stage_units[0]
        Ho, Wo = (x.shape[1] - 1) // stride + 1, (x.shape[2] - 1) // stride + 1
        out = torch.empty((B, Ho, Wo, oc), device=x.device, dtype=x.dtype)
        bounds = [(B * i) // nsplit for i in range(nsplit + 1)]
        self.last_stage_split[stage] = [hi - lo for lo, hi in zip(bounds, bounds[1:])]
        for lo, hi in zip(bounds, bounds[1:]):
            xs, ys = x[lo:hi], None
            for unit in stage_units:
                xs, ys = self._unit_hip(xs, unit, ys, sc_out=out[lo:hi] if unit[7] else None, inplace=self.inplace_expand, force_chain=True)
            if xs.data_ptr() != out[lo:hi].data_ptr():
                out[lo:hi].copy_(xs)
        return out

    def _unit_hip(self, x, unit, y_pre=None, sc_out=None, inplace=False, force_chain=False):
        """One bottleneck unit (resnet_v1_101_rcnn_base.py: branch1 | branch2a -> 2b -> 2c, + shortcut, ReLU) on the HIP kernels.
        y_pre: this unit's reduce output if the previous unit's chain kernel already produced it.  sc_out: where the projection
        shortcut of a first unit is written.  inplace: the expand kernel may write x_next over its shortcut operand.
        -> (x_next, reduce output of the NEXT unit | None)"""
        stage, nm, ic, mc, oc, stride, dil, proj = unit
        fuse_proj = (nm in self.chain_proj and sc_out is None and x.is_contiguous()
                     and (force_chain or ops.chain_worthwhile(x.numel() // x.shape[-1], mc)))
        if fuse_proj:
            sc = None                # res2a: the projection is computed inside the expand kernel
        elif proj:
            w, b, k = self.wp['res%s_branch1' % nm]
            sc = ops.conv2d_nhwc(x, w, b, ksize=k, stride=stride, out=sc_out)
        else:
            sc = x
        y = y_pre if y_pre is not None else self._hconv(x, 'res%s_branch2a' % nm, stride=stride, relu=True)
        if self.dcn and stage == 5:
            y = self._deform_2b(y.permute(0, 3, 1, 2), 'res%s_branch2b' % nm,
                                self._hconv(y, 'res%s_branch2b_offset' % nm, pad=2, dil=2, out_dtype=torch.float32).permute(0, 3, 1, 2))
            y = y.permute(0, 2, 3, 1)
        elif ('res%s_branch2b' % nm) in self.halo3:
            y = ops.conv3x3_halo(y.contiguous(), *self.halo3['res%s_branch2b' % nm], relu=True)
        else:
            y = self._hconv(y, 'res%s_branch2b' % nm, pad=dil, dil=dil, relu=True)
        if fuse_proj:
            if nm not in self.last_chain_units:
                self.last_chain_units.append(nm)
            return ops.bottleneck_chain_proj(y, x, *self.chain_proj[nm])
        ch = self.chain.get(nm)
        if ch is not None and not force_chain and not ops.chain_worthwhile(y.numel() // y.shape[-1], y.shape[-1]):
            ch = None            # small maps (B = 1, late stages at small B): the tiled convolution kernels fill the GPU better
        if ch is not None:       # expand + shortcut + ReLU and the next unit's reduce + ReLU in one pixel-wise kernel
            if nm not in self.last_chain_units:
                self.last_chain_units.append(nm)  # (a split stage, `inplace`, is only entered when every CU gets a 256-pixel set)
            return ops.bottleneck_chain(y, sc.contiguous(), *ch, inplace=inplace and sc.is_contiguous())
        return self._hconv(y, 'res%s_branch2c' % nm, relu=True, resid=sc), None     # relu(bn(conv) + shortcut)

    def _rpn_head(self, conv4):
        r = self._hconv(conv4, 'rpn_conv_3x3', pad=1, relu=True)
        rpn = self._hconv(r, 'rpn_out', out_dtype=torch.float32)
        na2 = self.w['rpn_cls_score'][0].shape[0]
        return rpn[..., :na2].permute(0, 3, 1, 2), rpn[..., na2:].permute(0, 3, 1, 2)

    def _side_stream(self):
        if getattr(self, '_side', None) is None:
            self._side = torch.cuda.Stream(device=self.device)
        return self._side

    def _deform_2b(self, y, name, offset):
        """DeformableConvolution(kernel 3x3, pad 2, dilate 2, num_deformable_group 4, no_bias) + BN + ReLU
        (SYM_DCN_RELNMS:700-707); y / offset logical NCHW."""
        w, b = self.wp_dcn[name]
        return ops.deformable_conv(y, offset, w, b, 3, 1, 2, 2, 4, relu=True)

    def _conv(self, x, name, stride=1, pad=0, dil=1, relu=False):
        w, b = self.w[name]
        y = F.conv2d(x, w, b, stride=stride, padding=pad, dilation=dil)
        return F.relu_(y) if relu else y

    def _c32(self, x, name, stride=1, pad=0, dil=1, relu=False, resid=None):
        w, b, k = self.wp32[name]
        return ops.conv2d_nhwc_f32(x, w, b, ksize=k, stride=stride, pad=pad, dil=dil, relu=relu, resid=resid)

    def _forward_hip32(self, data):
        """The float32 parity path on this repository's own kernels: every convolution on relnet_conv2d_nhwc_f32 (exact-fp32 MFMA, bias /
        shortcut / ReLU in the epilogue), pool1 on relnet_maxpool_nhwc_f32, NHWC float32 activations."""
        B, Cin, H, W = data.shape
        x = torch.zeros((B, H, W, 16), device=data.device, dtype=torch.float32)       # NHWC, channels zero-padded 3 -> 16
        x[..., :Cin] = data.permute(0, 2, 3, 1)
        x = self._c32(x, 'conv1', stride=2, pad=3, relu=True)
        x = ops.maxpool_nhwc_f32(x, 3, 2)
        conv4, ends = None, {}
        nchw = lambda t: t.permute(0, 3, 1, 2)
        for stage, nm, ic, mc, oc, stride, dil, proj in self.units:
            if stage == 5 and conv4 is None:
                conv4 = x
            if proj and stage > 2:
                ends[stage - 1] = x
            sc = self._c32(x, 'res%s_branch1' % nm, stride=stride) if proj else x
            y = self._c32(x, 'res%s_branch2a' % nm, stride=stride, relu=True)
            if self.dcn and stage == 5:
                off = self._c32(y, 'res%s_branch2b_offset' % nm, pad=2, dil=2)
                y = self._deform_2b(nchw(y), 'res%s_branch2b' % nm, nchw(off)).permute(0, 2, 3, 1)
            else:
                y = self._c32(y, 'res%s_branch2b' % nm, pad=dil, dil=dil, relu=True)
            x = self._c32(y, 'res%s_branch2c' % nm, relu=True, resid=sc)
        conv5 = x
        if self.frozen_only:
            return x
        if self.fpn:
            tops = {32: self._c32(conv5, 'fpn_ft32_1x1')}
            for lvl, src in ((16, ends[4]), (8, ends[3]), (4, ends[2])):
                tops[lvl] = ops.upsample2x_add_(self._c32(src, 'fpn_ft%d_1x1' % lvl), tops[lvl * 2])
            out = {'fpn_ft%d' % lvl: nchw(self._c32(tops[lvl], 'fpn_ft%d_3x3' % lvl, pad=1)) for lvl in (4, 8, 16, 32)}
            out.update(conv4=nchw(conv4), conv5=nchw(conv5))
            return out
        feat = self._c32(conv5, 'conv_new_1', relu=True)
        r = self._c32(conv4, 'rpn_conv_3x3', pad=1, relu=True)
        rpn = self._c32(r, 'rpn_out')
        na2 = self.w['rpn_cls_score'][0].shape[0]
        return dict(conv4=nchw(conv4), conv5=nchw(conv5), conv_new_1_relu=nchw(feat),
                    rpn_cls_score=nchw(rpn[..., :na2]), rpn_bbox_pred=nchw(rpn[..., na2:]))

    def forward(self, data, rpn_hook=None):
        """data [B,3,H,W] -> dict(conv4, conv5, conv_new_1_relu, rpn_cls_score, rpn_bbox_pred)
        (logical NCHW tensors; channels-last memory).

        rpn_hook (impl 'hip'): callable(rpn_cls_score, rpn_bbox_pred) -> anything, e.g. the proposal operator.  The RPN head
        and the hook then run on a SIDE stream as soon as conv4 exists, concurrently with res5 / conv_new_1 on the caller's
        stream (the proposal kernels are latency bound and occupy ~54 workgroups; res5 fills the rest of the GPU); the two
        streams join before this function returns and the hook's result is returned under the key 'rpn_hook'."""
        if self.impl == 'hip':
            return self._forward_hip(data, rpn_hook)
        if self.impl == 'hip32':
            return self._forward_hip32(data)
        x = data.to(self.dtype).contiguous(memory_format=self.mf)
        x = self._conv(x, 'conv1', stride=2, pad=3, relu=True)
        x = F.max_pool2d(x, kernel_size=3, stride=2, padding=0, ceil_mode=True)
        conv4 = None
        ends = {}
        for stage, nm, ic, mc, oc, stride, dil, proj in self.units:
            if stage == 5 and conv4 is None:
                conv4 = x
            if proj and stage > 2:
                ends[stage - 1] = x
            sc = self._conv(x, 'res%s_branch1' % nm, stride=stride) if proj else x
            y = self._conv(x, 'res%s_branch2a' % nm, stride=stride, relu=True)
            if self.dcn and stage == 5:
                off = self._conv(y, 'res%s_branch2b_offset' % nm, pad=2, dil=2).float()
                y = self._deform_2b(y, 'res%s_branch2b' % nm, off).contiguous(memory_format=self.mf)
            else:
                y = self._conv(y, 'res%s_branch2b' % nm, pad=dil, dil=dil, relu=True)
            y = self._conv(y, 'res%s_branch2c' % nm)
            x = F.relu_(y.add_(sc))
        conv5 = x
        if self.fpn:
            tops = {32: self._conv(conv5, 'fpn_ft32_1x1')}
            for lvl, src in ((16, ends[4]), (8, ends[3]), (4, ends[2])):
                tops[lvl] = F.interpolate(tops[lvl * 2], scale_factor=2, mode='nearest') + self._conv(src, 'fpn_ft%d_1x1' % lvl)
            out = {'fpn_ft%d' % lvl: self._conv(tops[lvl], 'fpn_ft%d_3x3' % lvl, pad=1) for lvl in (4, 8, 16, 32)}
            out.update(conv4=conv4, conv5=conv5)
            return out
        feat = self._conv(conv5, 'conv_new_1', relu=True)
        r = self._conv(conv4, 'rpn_conv_3x3', pad=1, relu=True)
        rpn = self._conv(r, 'rpn_out')
        na2 = self.w['rpn_cls_score'][0].shape[0]
        return dict(conv4=conv4, conv5=conv5, conv_new_1_relu=feat,
                    rpn_cls_score=rpn[:, :na2], rpn_bbox_pred=rpn[:, na2:])
