"""Executor of the `mx` facade: runs a symbolic graph built by the reference's own graph files on the GPU.

  * `infer_shapes`  -- one pass over the graph on torch `meta` tensors (no arithmetic): parameter shapes are
                       derived where an operator consumes a still-unknown variable (MXNet's InferShape).
  * `Executor`      -- `sym.bind(ctx, args, aux_states=...)`, `forward(is_train=False, **inputs)`, `outputs`.
                       Demand driven: only nodes the requested heads depend on are evaluated.
    Heavy operators go to librelnet_hip.so:
        Convolution (+ folded BatchNorm + ReLU + shortcut add, fused when the graph spells that chain)
                                   -> relnet_conv2d_nhwc / relnet_stem_conv7 (bf16) ; float32 executor: MIOpen
        FullyConnected, dot, batch_dot -> relnet_gemm_nt ;  ROIPooling -> relnet_roi_pool_fwd
        Pooling(max 3x3/2 'full')  -> relnet_stem_bias_relu_pool ;  Custom(op_type=...) -> operator_py registry
        the ~25-operator `attention_module_multi_head` sub-graph (SYM_REL:85-151 incl. the position
        matrix / embedding, :29-83)  -> geometry + projection + fused attention kernels (relation.py),
        after a bind-time numerical probe of the generic interpretation against the fused path.
    Light operators are torch tensor expressions on the device (mx/registry.py).
There is no CPU execution: binding anything but CUDA tensors raises.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import registry as R
from .registry import OPS, a_bool, a_int, a_tuple, a_str, a_float, a_get
from .ndarray import NDArray
from .. import ops as K
from .. import lib as _lib


# ---------------------------------------------------------------------------------------------------------
# shape inference
# ---------------------------------------------------------------------------------------------------------
def infer_shapes(sym, known):
    vals, var_shapes = {}, {}
    for node in sym._topo():
        if node.op == 'null':
            shp = known.get(node.name, R.a_tuple(node.attrs, '__shape__'))
            var_shapes[node.name] = tuple(shp) if shp is not None else None
            vals[(id(node), 0)] = R._meta(shp) if shp is not None else None
            continue
        d = OPS[node.op]
        ins = [vals[(id(s), i)] for s, i in node.inputs]
        if any(x is None for x in ins):
            names = d.inputs_for(node.attrs) if not d.variadic else [str(i) for i in range(len(ins))]
            if node.op == 'Custom':
                full = d.custom_params(node.attrs, [None if x is None else tuple(x.shape) for x in ins])
                want = dict(zip(names, [tuple(s) for s in full]))
            elif d.params is not None:
                want = d.params(node.attrs, {n: tuple(x.shape) for n, x in zip(names, ins) if x is not None})
            else:
                want = {}
            for k, (n, x) in enumerate(zip(names, ins)):
                if x is None:
                    if n not in want:
                        raise ValueError("cannot infer the shape of input %r of %s(%s)" % (n, node.op, node.name))
                    src = node.inputs[k][0]
                    var_shapes[src.name] = tuple(want[n])
                    vals[(id(src), 0)] = ins[k] = R._meta(want[n])
        out = d.fn(node.attrs, ctx={'device': 'meta'}) if getattr(d, 'no_input', False) else d.fn(node.attrs, *ins)
        outs = out if isinstance(out, (list, tuple)) else [out]
        for i, o in enumerate(outs):
            vals[(id(node), i)] = o
    return {'var': var_shapes, 'out': {k: tuple(v.shape) for k, v in vals.items() if v is not None}}


# ---------------------------------------------------------------------------------------------------------
# executor
# ---------------------------------------------------------------------------------------------------------
_LOWP_OK = {'Convolution', 'BatchNorm', 'Activation', 'broadcast_add', '_plus', 'elemwise_add', 'FullyConnected', 'ROIPooling', '_contrib_ROIAlign',
            'Pooling', 'slice_axis', '_contrib_DeformableConvolution', '_contrib_DeformablePSROIPooling', 'Flatten'}


def _t(x, device):
    if isinstance(x, NDArray):
        x = x.data
    if isinstance(x, np.ndarray):
        x = torch.as_tensor(x)
    return x.to(device)


class Executor(object):
    def __init__(self, sym, args, aux_states, ctx=None, dtype=torch.bfloat16, fuse=True, probe=True, device='cuda'):
        if not torch.cuda.is_available():
            raise _lib.RelnetError("the mx executor runs on the GPU only (HIP kernels; no CPU fallback)")
        _lib.load()
        self.sym, self.dtype, self.device = sym, dtype, device
        if isinstance(args, (list, tuple)):
            args = dict(zip(sym.list_arguments(), args))
        if isinstance(aux_states, (list, tuple)):
            aux_states = dict(zip(sym.list_auxiliary_states(), aux_states))
        self.arg_dict = {k: _t(v, device).float() for k, v in args.items()}
        self.aux_dict = {k: _t(v, device).float() for k, v in (aux_states or {}).items()}
        self.order = sym._topo()
        self.topo_index = {id(n): i for i, n in enumerate(self.order)}
        self.consumers = {}
        for n in self.order:
            for s, i in n.inputs:
                self.consumers.setdefault((id(s), i), []).append(n)
        self.head_keys = {(id(n), i) for n, i in sym.heads}
        self.cache = {}                 # per-node packed weights etc.
        self.override = {}              # id(node) -> (deps [(node, i)], fn(values...) -> outputs)
        self.fused_report = {'conv_chains': 0, 'attention_modules': 0, 'probe': []}
        self.conv_chain = {}            # id(last node of a fused conv chain) -> (conv, bn, relu, resid, last)
        self._parked = {}               # reduce outputs produced early by a block-boundary launch
        if fuse:
            if dtype == torch.bfloat16:
                self._fuse_conv_chains()
            self._fuse_attention(probe)
        self.outputs = []
        self.output_dict = {}

    # ---- helpers ----------------------------------------------------------------------------------------
    def _var(self, name):
        if name in self.arg_dict:
            return self.arg_dict[name]
        if name in self.aux_dict:
            return self.aux_dict[name]
        raise KeyError("argument %r was not bound" % name)

    def _sole_consumer(self, node, i=0, op=None):
        cs = self.consumers.get((id(node), i), [])
        if len(cs) != 1 or (id(node), i) in self.head_keys:
            return None
        return cs[0] if (op is None or cs[0].op in (op if isinstance(op, (tuple, set)) else (op,))) else None

    def _wants_fp32(self, node):
        if any((id(node), i) in self.head_keys for i in range(node.num_outputs)):
            return True
        for i in range(node.num_outputs):
            for c in self.consumers.get((id(node), i), []):
                if c.op not in _LOWP_OK:
                    return True
        return False

    # ---- conv + BN (+ReLU) (+shortcut add + ReLU) ---------------------------------------------------------
    def _fuse_conv_chains(self):
        for node in self.order:
            if node.op != 'Convolution' or a_int(node.attrs, 'num_group', 1) != 1:
                continue
            last, bn, relu, resid = node, None, False, None
            c = self._sole_consumer(last, 0, 'BatchNorm')
            if c is not None and a_bool(c.attrs, 'use_global_stats') and c.inputs[0][0] is last:
                bn, last = c, c
            c = self._sole_consumer(last, 0, 'Activation')
            if c is not None and a_str(c.attrs, 'act_type') == 'relu':
                relu, last = True, c
            else:
                c = self._sole_consumer(last, 0, ('broadcast_add', '_plus', 'elemwise_add'))
                if c is not None:
                    other = [h for h in c.inputs if h[0] is not last]
                    c2 = self._sole_consumer(c, 0, 'Activation')
                    # the shortcut operand must already exist when this chain runs (it is the EARLIER operand of the add):
                    # res2a = branch1 + branch2c fuses into branch2c's convolution, with branch1's conv+BN as the residual
                    if (len(other) == 1 and c2 is not None and a_str(c2.attrs, 'act_type') == 'relu'
                            and self.topo_index[id(other[0][0])] < self.topo_index[id(node)]):
                        resid, relu, last = other[0], True, c2
            if last is node and bn is None:
                continue
            deps = [node.inputs[0]] + ([resid] if resid is not None else [])
            self.override[id(last)] = (deps, self._conv_runner(node, bn, relu, resid is not None, last))
            self.conv_chain[id(last)] = (node, bn, relu, resid, last)
            self.fused_report['conv_chains'] += 1
        self._fuse_block_boundaries()

    def _fuse_block_boundaries(self):
        """expand (1x1, 4 mid outputs, + shortcut + ReLU) followed by the next unit's reduce (1x1 -> mid, + ReLU), mid in
        ops.CHAIN_MIDS: both run as ONE relnet_bottleneck_chain launch.  The expand node's runner computes both tensors and
        parks the reduce output for the reduce node, which is evaluated later in topological order."""
        is1x1 = lambda n: (a_tuple(n.attrs, 'kernel') == (1, 1) and a_tuple(n.attrs, 'stride', (1, 1)) == (1, 1)
                           and a_tuple(n.attrs, 'pad', (0, 0)) == (0, 0))
        for lid, (node, bn, relu, resid, last) in list(self.conv_chain.items()):
            nf = a_int(node.attrs, 'num_filter')
            if resid is None or not relu or not is1x1(node) or nf % 4 or (nf // 4) not in K.CHAIN_MIDS or self._wants_fp32(last):
                continue
            mid = nf // 4
            wshape = lambda n: self._var(n.inputs[OPS['Convolution'].inputs_for(n.attrs).index('weight')][0].name).shape
            if wshape(node)[1] != mid:
                continue
            nxt = [c for c in self.consumers.get((id(last), 0), []) if c.op == 'Convolution' and c.inputs[0][0] is last]
            nxt = [c for c in nxt if is1x1(c) and a_int(c.attrs, 'num_filter') == mid and a_int(c.attrs, 'num_group', 1) == 1]
            rc = [v for v in self.conv_chain.values() if len(nxt) == 1 and v[0] is nxt[0]]
            if not rc or rc[0][3] is not None or not rc[0][2] or self._wants_fp32(rc[0][4]) or wshape(rc[0][0])[1] != nf:
                # last unit of a stage (or a consumer the kernel does not cover): expand + shortcut + ReLU only
                self.override[lid] = (self.override[lid][0], self._chain_runner(node, bn, None, None, None, self.override[lid][1]))
                self.fused_report['block_boundaries'] = self.fused_report.get('block_boundaries', 0) + 1
                continue
            rnode, rbn, _, _, rlast = rc[0]
            self.override[lid] = (self.override[lid][0], self._chain_runner(node, bn, rnode, rbn, id(rlast), self.override[lid][1]))
            plain = self.override[id(rlast)][1]
            self.override[id(rlast)] = (self.override[id(rlast)][0] + [(last, 0)],
                                        lambda x, xn_dep, plain=plain, key=id(rlast): self._parked.pop(key) if key in self._parked else plain(x))
            self.fused_report['block_boundaries'] = self.fused_report.get('block_boundaries', 0) + 1

    def _chain_runner(self, node, bn, rnode, rbn, rkey, plain):
        def run(x, resid):
            key = ('chain', id(node))
            if key not in self.cache:
                w3, b3 = self._conv_weights(node, bn)
                w3f = K.pack_w_frag(K.pack_conv_weight(w3, torch.bfloat16, self.device))
                if rnode is None:
                    self.cache[key] = (w3f, None, b3, None)
                else:
                    w1, b1 = self._conv_weights(rnode, rbn)
                    self.cache[key] = (w3f, K.pack_chain_w1(K.pack_conv_weight(w1, torch.bfloat16, self.device)), b3, b1)
            nhwc = lambda t: t.permute(0, 2, 3, 1).to(torch.bfloat16).contiguous()
            if not K.chain_worthwhile(x.numel() // x.shape[1], x.shape[1]):      # same rule as Backbone (small maps)
                return plain(x, resid)
            xn, m1 = K.bottleneck_chain(nhwc(x), nhwc(resid), *self.cache[key])
            if m1 is not None:
                self._parked[rkey] = m1.permute(0, 3, 1, 2)
            return xn.permute(0, 3, 1, 2)
        return run

    def _depends_on(self, a, b):
        seen, stack = set(), [a]
        while stack:
            n = stack.pop()
            if n is b:
                return True
            if id(n) in seen:
                continue
            seen.add(id(n))
            stack.extend(s for s, _ in n.inputs)
        return False

    def _conv_weights(self, node, bn):
        key = ('conv', id(node))
        if key not in self.cache:
            names = OPS['Convolution'].inputs_for(node.attrs)
            w = self._var(node.inputs[names.index('weight')][0].name)
            b = self._var(node.inputs[names.index('bias')][0].name) if 'bias' in names else torch.zeros(w.shape[0], device=self.device)
            if bn is not None:
                g, be, mu, var = (self._var(bn.inputs[j][0].name) for j in (1, 2, 3, 4))
                if a_bool(bn.attrs, 'fix_gamma', True):
                    g = torch.ones_like(g)
                s = g.double() / torch.sqrt(var.double() + a_float(bn.attrs, 'eps', 1e-3))
                b = (be.double() + (b.double() - mu.double()) * s).float()
                w = (w.double() * s.view(-1, 1, 1, 1)).float()
            self.cache[key] = (w, b.float().contiguous())
        return self.cache[key]

    def _conv_runner(self, node, bn, relu, has_resid, last):
        a = node.attrs
        k, s, d, p = a_tuple(a, 'kernel'), a_tuple(a, 'stride', (1, 1)), a_tuple(a, 'dilate', (1, 1)), a_tuple(a, 'pad', (0, 0))
        out32 = self._wants_fp32(last)

        def run(x, resid=None):
            w, b = self._conv_weights(node, bn)
            return self._conv_bf16(node, x, w, b, k, s, d, p, relu, resid, out32)
        return run

    def _conv_bf16(self, node, x, w, b, k, s, d, p, relu, resid, out32):
        cin = w.shape[1]
        if cin == 3 and k == (7, 7) and s == (2, 2) and p == (3, 3):                   # conv1 (SYM_BASE:30-31)
            key = ('stem', id(node))
            if key not in self.cache:
                self.cache[key] = K.pack_stem_weight(w, torch.bfloat16, self.device)
            y = K.stem_conv7(x.float().contiguous(), self.cache[key], b, relu=relu)
            assert resid is None
            return y.permute(0, 3, 1, 2)
        if cin % 64 or k[0] != k[1] or s[0] != s[1] or d[0] != d[1] or p[0] != p[1]:
            raise NotImplementedError("Convolution %s: Cin=%d kernel=%s stride=%s is outside the implicit-GEMM kernel's domain "
                                      "(Cin %% 64 == 0, square geometry)" % (node.name, cin, k, s))
        key = ('packed', id(node))
        if key not in self.cache:
            self.cache[key] = K.pack_conv_weight(w, torch.bfloat16, self.device)
        xn = x.permute(0, 2, 3, 1)
        if xn.dtype != torch.bfloat16 or not xn.is_contiguous():
            xn = xn.to(torch.bfloat16).contiguous()
        odt = torch.float32 if out32 else torch.bfloat16
        if (k == (3, 3) and s == (1, 1) and d == (1, 1) and p == (1, 1) and cin in K.HALO3_CHANNELS and w.shape[0] == cin
                and resid is None and not out32):
            fkey = ('halo3', id(node))                    # 64- / 256-channel 3x3 (res2 / res4 branch2b): halo-resident kernels
            if fkey not in self.cache:
                self.cache[fkey] = K.pack_w_frag(self.cache[key], panel_only=False)
            return K.conv3x3_halo(xn, self.cache[fkey], b, relu=relu).permute(0, 3, 1, 2)
        rn = None
        if resid is not None:
            rn = resid.permute(0, 2, 3, 1)
            if rn.dtype != odt or not rn.is_contiguous():
                rn = rn.to(odt).contiguous()
        y = K.conv2d_nhwc(xn, self.cache[key], b, ksize=k[0], stride=s[0], pad=p[0], dil=d[0], relu=relu, resid=rn, out_dtype=odt)
        return y.permute(0, 3, 1, 2)

    # ---- attention module ---------------------------------------------------------------------------------
    def _fuse_attention(self, probe):
        from .. import relation
        for node in self.order:
            m = self._match_attention(node)
            if m is None:
                continue
            params = {}
            for canon, var in m['params'].items():
                params[canon] = self._var(var)
            packed = relation.RelationParams(params, 1, self.dtype, self.device)
            nongt = m['nongt_dim']

            def run(feat, rois, packed=packed, nongt=nongt):
                return relation.attention_module_multi_head(feat.to(self.dtype), rois.float().contiguous(), None, nongt_dim=nongt,
                                                            dtype=self.dtype, packed=packed)
            if probe:
                err = self._probe_attention(node, m, run)
                self.fused_report['probe'].append((node.name, err))
                tol = 3e-2 if self.dtype == torch.bfloat16 else 2e-4
                if not (err <= tol):
                    raise _lib.RelnetError("attention sub-graph ending in %s does not compute what the fused relation kernels do "
                                           "(probe rel. error %.3g > %.3g)" % (node.name, err, tol))
            self.override[id(m['out'])] = ([m['feat'], m['rois']], run)
            self.fused_report['attention_modules'] += 1

    @staticmethod
    def _src(node, op, idx=0, **attr_checks):
        s = node.inputs[idx][0]
        if s.op != op:
            return None
        return s

    def _match_attention(self, L):
        """Structural signature of SYM_REL:104-150 ending in the grouped 1x1 `linear_out` convolution."""
        if L.op != 'Convolution' or a_int(L.attrs, 'num_group', 1) != 16 or a_tuple(L.attrs, 'kernel') != (1, 1) or a_int(L.attrs, 'num_filter') != 1024:
            return None
        try:
            r1 = self._src(L, 'Reshape'); dot = self._src(r1, 'dot'); r2 = self._src(dot, 'Reshape'); sm = self._src(r2, 'softmax')
            plus = self._src(sm, '_plus')
            lg = self._src(plus, 'log', 0); mx_ = self._src(lg, '_maximum_scalar'); aw_t = self._src(mx_, 'transpose')
            aw_r = self._src(aw_t, 'Reshape'); act = self._src(aw_r, 'Activation'); pf = self._src(act, 'FullyConnected')
            pe_r = self._src(pf, 'Reshape')
            at = self._src(plus, 'transpose', 1); ms = self._src(at, '_mul_scalar'); bd = self._src(ms, 'batch_dot')
            q = self._src(self._src(self._src(bd, 'transpose', 0), 'Reshape'), 'FullyConnected')
            kk = self._src(self._src(self._src(bd, 'transpose', 1), 'Reshape'), 'FullyConnected')
            ng = self._src(kk, 'slice_axis')
        except AttributeError:
            return None
        if None in (q, kk, ng) or dot.inputs[1][0] is not ng or ng.inputs[0] != q.inputs[0]:
            return None
        if a_int(sm.attrs, 'axis', -1) != 2 or a_int(ng.attrs, 'axis') != 0 or a_int(ng.attrs, 'begin', 0) != 0:
            return None
        if abs(a_float(ms.attrs, 'scalar') - 0.125) > 1e-7 or abs(a_float(mx_.attrs, 'scalar') - 1e-6) > 1e-12:
            return None
        # the position embedding: walk back to the `slice_axis(rois, axis=1, begin=1)` that feeds extract_position_matrix
        pe = pe_r.inputs[0][0]
        rois_head, stack, seen = None, [pe], set()
        while stack:
            n = stack.pop()
            if id(n) in seen:
                continue
            seen.add(id(n))
            if n.op == 'slice_axis' and a_int(n.attrs, 'axis') == 1 and a_int(n.attrs, 'begin', 0) == 1:
                if rois_head is not None and rois_head != n.inputs[0]:
                    return None
                rois_head = n.inputs[0]
                continue
            if n.op == 'null' or OPS[n.op].heavy:
                return None                                  # embedding must be a pure function of the rois
            stack.extend(s for s, _ in n.inputs)
        if rois_head is None:
            return None
        out = L
        c = self._sole_consumer(L, 0, 'Reshape')
        if c is not None and a_tuple(c.attrs, 'shape') == (0, 0):
            out = c
        names = lambda n: [h[0].name for h in n.inputs[1:]]
        (wq, bq), (wk, bk), (wp, bp), (wo, bo) = names(q), names(kk), names(pf), names(L)
        return dict(out=out, feat=q.inputs[0], rois=rois_head, nongt_dim=a_int(ng.attrs, 'end'), pe=pe_r.inputs[0],
                    params={'query_1_weight': wq, 'query_1_bias': bq, 'key_1_weight': wk, 'key_1_bias': bk,
                            'pair_pos_fc1_1_weight': wp, 'pair_pos_fc1_1_bias': bp, 'linear_out_1_weight': wo, 'linear_out_1_bias': bo})

    def _probe_attention(self, out_node, m, fused_run):
        """Generic (operator by operator) evaluation of the matched sub-graph vs the fused kernels on seeded inputs."""
        g = torch.Generator().manual_seed(1234)
        n = max(m['nongt_dim'], 1) + 4
        x1 = torch.rand(n, generator=g) * 800; y1 = torch.rand(n, generator=g) * 450
        w = torch.rand(n, generator=g) * 180 + 16; h = torch.rand(n, generator=g) * 130 + 16
        rois = torch.stack([torch.zeros(n), x1, y1, x1 + w, y1 + h], 1).to(self.device)
        feat = torch.randn(n, 1024, generator=g).to(self.device)
        saved = self.override
        self.override = {}
        try:
            want = self._evaluate([(m['out'], 0)], preset={m['feat']: feat, m['rois']: rois}, dtype=torch.float32)[0]
        finally:
            self.override = saved
        got = fused_run(feat, rois).float()
        return float((got - want).abs().max() / want.abs().max())

    # ---- evaluation ---------------------------------------------------------------------------------------
    def forward(self, is_train=False, **kwargs):
        for k, v in kwargs.items():
            self.arg_dict[k] = _t(v, self.device).float()
        with torch.no_grad():
            outs = self._evaluate(self.sym.heads)
        self.outputs = [NDArray(o.float() if o.is_floating_point() else o) for o in outs]
        self.output_dict = dict(zip(self.sym.list_outputs(), self.outputs))
        return self.outputs

    def _evaluate(self, heads, preset=None, dtype=None):
        dtype = dtype or self.dtype
        preset = {(id(n), i): v for (n, i), v in (preset or {}).items()}
        # nodes needed, honouring fused overrides
        need, stack = set(), [n for n, _ in heads]
        deps_of = {}
        while stack:
            n = stack.pop()
            if id(n) in need:
                continue
            need.add(id(n))
            if all((id(n), i) in preset for i in range(max(n.num_outputs, 1))):
                deps_of[id(n)] = []
                continue
            deps = self.override[id(n)][0] if id(n) in self.override else n.inputs
            deps_of[id(n)] = deps
            stack.extend(s for s, _ in deps)
        vals = dict(preset)
        for n in self.order:
            if id(n) not in need or (id(n), 0) in vals:
                continue
            ins = [vals[(id(s), i)] for s, i in deps_of[id(n)]]
            if n.op == 'null':
                out = self._var(n.name)
            elif id(n) in self.override:
                out = self.override[id(n)][1](*ins)
            else:
                out = self._run_op(n, ins, dtype)
            outs = out if isinstance(out, (list, tuple)) else [out]
            for i, o in enumerate(outs):
                vals[(id(n), i)] = o
        return [vals[(id(n), i)] for n, i in heads]

    def _gemm(self, a2d, w2d, bias, dtype, out_dtype, relu=False):
        """a [M,K] x w [N,K]^T on relnet_gemm_nt; K zero-padded to the kernel's granularity."""
        gran = 64 if dtype == torch.bfloat16 else 16
        a2d, w2d = a2d.to(dtype), w2d.to(dtype)
        Kd = a2d.shape[1]
        if Kd % gran:
            pad = gran - Kd % gran
            a2d, w2d = F.pad(a2d, (0, pad)), F.pad(w2d, (0, pad))
        return K.gemm_nt(a2d.contiguous(), w2d.contiguous(), None if bias is None else bias.float().contiguous(), relu=relu, out_dtype=out_dtype)

    def _run_op(self, n, ins, dtype):
        d, a = OPS[n.op], n.attrs
        if getattr(d, 'no_input', False):
            return d.fn(a, ctx={'device': self.device})
        if not d.heavy:
            return d.fn(a, *ins)
        odt = torch.float32 if (dtype == torch.float32 or self._wants_fp32(n)) else torch.bfloat16
        if n.op == 'Convolution':
            x, w = ins[0], ins[1]
            b = ins[2] if len(ins) > 2 else torch.zeros(w.shape[0], device=self.device)
            k, s, dl, p = a_tuple(a, 'kernel'), a_tuple(a, 'stride', (1, 1)), a_tuple(a, 'dilate', (1, 1)), a_tuple(a, 'pad', (0, 0))
            g = a_int(a, 'num_group', 1)
            if dtype == torch.float32 or g != 1:
                # float32 parity executor: library convolution (MIOpen), as the detector's float32 path; grouped 1x1
                # convolutions only occur inside attention modules (fused) -- as a per-group GEMM here
                if g != 1 and k == (1, 1):
                    xg = x.reshape(x.shape[0], g, -1).float()
                    wg = w.reshape(g, w.shape[0] // g, -1).float()
                    outs = [self._gemm(xg[:, i], wg[i], b[i * wg.shape[1]:(i + 1) * wg.shape[1]], torch.float32, torch.float32) for i in range(g)]
                    return torch.cat(outs, 1).reshape(x.shape[0], -1, 1, 1)
                return F.conv2d(x.float(), w, b, stride=s, padding=p, dilation=dl, groups=g)
            return self._conv_bf16(n, x, w, b.float().contiguous(), k, s, dl, p, False, None, odt == torch.float32)
        if n.op == 'FullyConnected':
            x, w = ins[0], ins[1]
            b = ins[2] if len(ins) > 2 else None
            x2 = x.reshape(x.shape[0], -1) if a_bool(a, 'flatten', True) else x.reshape(-1, x.shape[-1])
            y = self._gemm(x2, w, b, dtype, odt)
            return y if a_bool(a, 'flatten', True) else y.reshape(tuple(x.shape[:-1]) + (w.shape[0],))
        if n.op == 'dot':
            x, y = ins
            x = x.t() if a_bool(a, 'transpose_a') else x
            wt = y if a_bool(a, 'transpose_b') else y.t()
            return self._gemm(x.contiguous(), wt.contiguous(), None, dtype, odt)
        if n.op == 'batch_dot':
            x, y = ins
            x = x.transpose(1, 2) if a_bool(a, 'transpose_a') else x
            wt = y if a_bool(a, 'transpose_b') else y.transpose(1, 2)
            return torch.stack([self._gemm(x[i].contiguous(), wt[i].contiguous(), None, dtype, odt) for i in range(x.shape[0])])
        if n.op == 'ROIPooling':
            x, rois = ins
            ps = a_tuple(a, 'pooled_size')
            return K.roi_pool(x if x.dtype in (torch.float32, torch.bfloat16) else x.float(), rois.float().contiguous(), ps,
                              a_float(a, 'spatial_scale'))
        if n.op == '_contrib_ROIAlign':
            x, rois = ins
            return K.roi_align(x if x.dtype in (torch.float32, torch.bfloat16) else x.float(), rois.float().contiguous(), a_tuple(a, 'pooled_size'),
                               a_float(a, 'spatial_scale'), a_int(a, 'sample_ratio', -1), a_bool(a, 'aligned'))
        if n.op == 'Pooling':
            x = ins[0]
            k, s, p = a_tuple(a, 'kernel'), a_tuple(a, 'stride', (1, 1)), a_tuple(a, 'pad', (0, 0))
            full = a_str(a, 'pooling_convention', 'valid') == 'full'
            src = n.inputs[0][0]
            relu_in = (src.op == 'Activation' and a_str(src.attrs, 'act_type') == 'relu')
            if (a_str(a, 'pool_type', 'max') == 'max' and k == (3, 3) and s == (2, 2) and p == (0, 0) and full and relu_in
                    and x.dtype == torch.bfloat16 and x.permute(0, 2, 3, 1).is_contiguous()):
                key = ('zero_bias', x.shape[1])
                if key not in self.cache:
                    self.cache[key] = torch.zeros(x.shape[1], device=self.device)
                return K.stem_bias_relu_pool(x.permute(0, 2, 3, 1), self.cache[key]).permute(0, 3, 1, 2)
            if a_bool(a, 'global_pool'):
                return x.float().mean((2, 3), keepdim=True) if a_str(a, 'pool_type') == 'avg' else x.float().amax((2, 3), keepdim=True)
            fn = F.max_pool2d if a_str(a, 'pool_type', 'max') == 'max' else F.avg_pool2d
            return fn(x.float(), k, s, p, ceil_mode=full)
        if n.op == 'Custom':
            prop = R.custom_prop(a)
            in_data = [x.float().contiguous() for x in ins]
            shapes = prop.infer_shape([tuple(x.shape) for x in in_data])
            out_data = [torch.empty(sh, device=self.device, dtype=torch.float32) for sh in shapes[1]]
            op = prop.create_operator(None, [tuple(x.shape) for x in in_data], None)
            op.forward(False, ['write'] * len(out_data), in_data, out_data, [])
            return out_data if len(out_data) > 1 else out_data[0]
        if n.op == '_contrib_DeformableConvolution':
            from .. import operator_cxx
            names = OPS[n.op].inputs_for(a)
            kw = dict(zip(names, ins))
            return operator_cxx.contrib.DeformableConvolution(**kw, **{k_: a_get(a, k_) for k_ in a if k_ != 'name'})
        if n.op == '_contrib_DeformablePSROIPooling':
            from .. import operator_cxx
            names = OPS[n.op].inputs_for(a)
            kw = dict(zip(names, ins))
            return operator_cxx.contrib.DeformablePSROIPooling(**kw, **{k_: a_get(a, k_) for k_ in a if k_ != 'name'})
        raise NotImplementedError("operator %s has no device implementation in this executor" % n.op)
