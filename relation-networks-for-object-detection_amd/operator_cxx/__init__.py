"""Mirror of the reference's C++ operator interface for the DCN configuration
(relation_rcnn/operator_cxx: `_contrib_DeformableConvolution`, `_contrib_DeformablePSROIPooling`).

The reference registers legacy MXNet operators: a `*Param` struct, an `OperatorProperty`
(ListArguments / ListOutputs / InferShape) and an `Operator` with `Forward(ctx, in_data, req,
out_data, aux)`.  The same three pieces exist here over torch device tensors, with the reference's
parameter names, defaults, argument order, shape rules and failure conditions; `Forward` launches the
HIP kernels of csrc/deform.hip + the GEMM kernel.  `contrib` carries the Python call surface the symbol
files use (`mx.contrib.sym.DeformableConvolution(...)`, SYM_DCN_RELNMS:702-704, 1073-1080).

Tensors are the reference's: NCHW, fp32 (or bf16 / channels-last memory format for the throughput
path -- the kernels take explicit strides).  `Backward(ctx, out_grad, in_data, out_data, req, in_grad, aux)` follows
deformable_convolution-inl.h:145-237 / deformable_psroi_pooling-inl.h:97-140 (`req` per input: 'null' | 'write' | 'add';
the data / offset / trans gradients come from relnet_deformable_col2im / relnet_deformable_psroi_pool_bwd).
A C++ binding of the same classes over the C-ABI is include/relnet_operator_cxx.hpp.
"""
import torch

from .. import ops

kWriteTo, kNullOp, kAddTo, kWriteInplace = 'write', 'null', 'add', 'inplace'


def _assign(dst, req, src):
    """mshadow ASSIGN_DISPATCH / Assign: kNullOp leaves dst alone, kAddTo accumulates, kWriteTo / kWriteInplace overwrite."""
    if req == kNullOp:
        return
    src = src.to(dst.dtype).reshape(dst.shape)
    if req == kAddTo:
        dst.add_(src)
    elif req in (kWriteTo, kWriteInplace):
        dst.copy_(src)
    else:
        raise ValueError('unknown OpReqType %r' % (req,))


def _shape2(v, default):
    if v is None or (isinstance(v, (tuple, list)) and len(v) == 0):
        return default
    if isinstance(v, str):                      # MXNet attribute strings: '(3, 3)'
        v = tuple(int(x) for x in v.strip('()[] ').split(',') if x.strip())
    if isinstance(v, int):
        return (v, v)
    if len(v) != 2:
        raise ValueError("only 2-D deformable convolution is supported (deformable_convolution-inl.h:346-348), got %r" % (v,))
    return (int(v[0]), int(v[1]))


def _bool(v):
    return v if isinstance(v, bool) else str(v).lower() in ('1', 'true')


class DeformableConvolutionParam(object):
    """deformable_convolution-inl.h:39-76 (kernel required; stride/dilate default (1,1), pad (0,0))."""

    def __init__(self, kernel, num_filter, stride=None, dilate=None, pad=None, num_group=1,
                 num_deformable_group=1, workspace=1024, no_bias=False, layout=None):
        self.kernel = _shape2(kernel, None)
        if self.kernel is None:
            raise ValueError("DeformableConvolution: kernel is required")
        self.stride = _shape2(stride, (1, 1))
        self.dilate = _shape2(dilate, (1, 1))
        self.pad = _shape2(pad, (0, 0))
        self.num_filter = int(num_filter)
        if not 1 <= self.num_filter <= 100000:
            raise ValueError("num_filter out of range [1, 100000]")
        self.num_group = int(num_group)
        self.num_deformable_group = int(num_deformable_group)
        self.workspace = int(workspace)
        self.no_bias = _bool(no_bias)
        if layout not in (None, 'NCHW'):
            raise ValueError("DeformableConvolution: only the NCHW layout is supported")
        self.layout = 'NCHW'


class DeformableConvolutionProp(object):
    """OperatorProperty of `_contrib_DeformableConvolution` (deformable_convolution-inl.h:294-470)."""

    def __init__(self, **kwargs):
        self.param_ = DeformableConvolutionParam(**kwargs)

    def ListArguments(self):
        return ['data', 'offset', 'weight'] if self.param_.no_bias else ['data', 'offset', 'weight', 'bias']

    def ListOutputs(self):
        return ['output']

    def InferShape(self, in_shape):
        """in_shape: [data, offset, weight(, bias)] (weight / bias may be None) -> (in_shapes, [out_shape])."""
        p = self.param_
        expected = 3 if p.no_bias else 4
        if len(in_shape) != expected:
            raise ValueError("Input:[data, offset, weight%s]" % ('' if p.no_bias else ', bias'))
        dshape, oshape = tuple(in_shape[0]), tuple(in_shape[1])
        if len(dshape) != 4 or len(oshape) != 4:
            raise ValueError("Input data / offset should be 4D in batch-num_filter-y-x")
        if dshape[1] % p.num_group or dshape[1] % p.num_deformable_group or p.num_filter % p.num_group:
            raise ValueError("input channels / num_filter must divide num_group and num_deformable_group")
        wshape = (p.num_filter, dshape[1] // p.num_group, p.kernel[0], p.kernel[1])
        ho = (dshape[2] + 2 * p.pad[0] - (p.dilate[0] * (p.kernel[0] - 1) + 1)) // p.stride[0] + 1
        wo = (dshape[3] + 2 * p.pad[1] - (p.dilate[1] * (p.kernel[1] - 1) + 1)) // p.stride[1] + 1
        out = (dshape[0], p.num_filter, ho, wo)
        if out[1] % p.num_deformable_group:
            raise ValueError("output num_filter must divide deformable group size")
        if (oshape[2], oshape[3]) != (ho, wo):
            raise ValueError("output height / width must equal the offset map's")
        if oshape[1] % (p.kernel[0] * p.kernel[1]) or oshape[1] // (2 * p.kernel[0] * p.kernel[1]) != p.num_deformable_group:
            raise ValueError("offset filter must divide deformable group size")
        shapes = [dshape, oshape, wshape] + ([] if p.no_bias else [(p.num_filter,)])
        return shapes, [out]

    def InferType(self, in_type):
        """deformable_convolution-inl.h:419-437: one uniform dtype; unspecified (None) entries take the first input's."""
        if not in_type or in_type[0] is None:
            raise ValueError("First input must have specified type")
        dt = in_type[0]
        for i, t in enumerate(in_type):
            if t is not None and t != dt:
                raise ValueError("This layer requires uniform type. Expected %s v.s. given %s at %s" % (dt, t, self.ListArguments()[i]))
        return [dt] * len(in_type), [dt]

    def TypeString(self):
        return '_contrib_DeformableConvolution'

    def DeclareBackwardDependency(self, out_grad, in_data, out_data):
        return [out_grad[0], in_data[0], in_data[1], in_data[2]]          # :448-453: kOut grad, kData, kOffset, kWeight

    def CreateOperatorEx(self, ctx=None, in_shape=None, in_type=None):
        return DeformableConvolutionOp(self.param_)


class DeformableConvolutionOp(object):
    """DeformableConvolutionOp::Forward (deformable_convolution-inl.h:91-143)."""

    def __init__(self, param):
        self.param_ = param
        if param.num_group != 1:
            raise NotImplementedError("DeformableConvolution with num_group > 1 is not built (unused by the reference graphs)")
        self._packed = None

    def _pack(self, weight):
        key = (weight.data_ptr(), weight._version, weight.dtype)
        if self._packed is None or self._packed[0] != key:
            self._packed = (key, ops.pack_conv_weight(weight, dtype=weight.dtype, device=weight.device))
        return self._packed[1]

    def Forward(self, ctx, in_data, req, out_data, aux_args=None):
        p = self.param_
        if req[0] != kWriteTo:
            raise ValueError("DeformableConvolution: req[kOut] must be kWriteTo (deformable_convolution-inl.h:98)")
        if len(in_data) != (3 if p.no_bias else 4) or len(out_data) != 1:
            raise ValueError("DeformableConvolution: wrong number of inputs / outputs")
        data, offset, weight = in_data[0], in_data[1], in_data[2]
        bias = None if p.no_bias else in_data[3].float()
        y = ops.deformable_conv(data, offset.float(), self._pack(weight), bias, p.kernel, p.stride, p.dilate, p.pad,
                                p.num_deformable_group, out_dtype=out_data[0].dtype)
        out_data[0].copy_(y)

    def Backward(self, ctx, out_grad, in_data, out_data, req, in_grad, aux_args=None):
        """DeformableConvolutionOp::Backward (deformable_convolution-inl.h:145-237): in_grad = [d data, d offset, d weight(, d bias)]
        NCHW / OIHW like in_data; the data gradient starts from zero (`data_grad = 0`, :189-190) unless req is 'add'."""
        p = self.param_
        expected = 3 if p.no_bias else 4
        if len(out_grad) != 1 or len(in_data) != expected or len(in_grad) != expected or len(req) != expected:
            raise ValueError("DeformableConvolution.Backward: wrong number of out_grad / in_data / in_grad / req entries")
        data, offset, weight = in_data[0], in_data[1], in_data[2]
        if not weight.is_contiguous():
            raise ValueError("DeformableConvolution.Backward: weight must be contiguous (:158)")
        dy = out_grad[0]
        cdt = data.dtype if data.dtype in (torch.float32, torch.bfloat16) else torch.float32
        dy_cl = dy.to(cdt).contiguous(memory_format=torch.channels_last)       # [B,Ho,Wo,Cout] memory the kernels read
        gdata, goff, gw = ops.deformable_conv_bwd(data.to(cdt), offset.float(), self._pack(weight.to(cdt)), dy_cl, p.kernel, p.stride,
                                                  p.dilate, p.pad, p.num_deformable_group)
        kh, kw = p.kernel
        _assign(in_grad[0], req[0], gdata.permute(0, 3, 1, 2))
        _assign(in_grad[1], req[1], goff.permute(0, 3, 1, 2))
        _assign(in_grad[2], req[2], gw.view(p.num_filter, kh, kw, -1).permute(0, 3, 1, 2))
        if not p.no_bias:
            _assign(in_grad[3], req[3], dy.float().sum((0, 2, 3)))                # sumall_except_dim<1>, :228-233


class DeformablePSROIPoolingParam(object):
    """deformable_psroi_pooling-inl.h:32-55."""

    def __init__(self, spatial_scale, output_dim, group_size, pooled_size, part_size=0, sample_per_part=1,
                 trans_std=0.0, no_trans=False):
        self.spatial_scale = float(spatial_scale)
        if not 0.0 <= self.spatial_scale <= 1.0:
            raise ValueError("spatial_scale out of range [0, 1]")
        self.output_dim, self.group_size, self.pooled_size = int(output_dim), int(group_size), int(pooled_size)
        self.part_size, self.sample_per_part = int(part_size), int(sample_per_part)
        self.trans_std = float(trans_std)
        if not 0.0 <= self.trans_std <= 1.0:
            raise ValueError("trans_std out of range [0, 1]")
        self.no_trans = _bool(no_trans)


class DeformablePSROIPoolingProp(object):
    """OperatorProperty of `_contrib_DeformablePSROIPooling` (deformable_psroi_pooling-inl.h:153-270)."""

    def __init__(self, **kwargs):
        self.param_ = DeformablePSROIPoolingParam(**kwargs)

    def ListArguments(self):
        return ['data', 'rois'] if self.param_.no_trans else ['data', 'rois', 'trans']

    def ListOutputs(self):
        return ['output', 'top_count']

    def NumVisibleOutputs(self):
        return 1

    def InferShape(self, in_shape):
        p = self.param_
        if len(in_shape) != (2 if p.no_trans else 3):
            raise ValueError("Input:[data, rois%s]" % ('' if p.no_trans else ', trans'))
        dshape, bshape = tuple(in_shape[0]), tuple(in_shape[1])
        if len(dshape) != 4:
            raise ValueError("data should be a 4D tensor")
        if len(bshape) != 2 or bshape[1] != 5:
            raise ValueError("bbox should be a 2D tensor of shape [batch, 5]")
        out = (bshape[0], p.output_dim, p.pooled_size, p.pooled_size)
        return list(in_shape), [out, out]

    def TypeString(self):
        return '_contrib_DeformablePSROIPooling'

    def DeclareBackwardDependency(self, out_grad, in_data, out_data):
        # deformable_psroi_pooling-inl.h:243-255: kOut grad, kData, kBox(, kTrans), kTopCount
        return [out_grad[0]] + list(in_data[:2 if self.param_.no_trans else 3]) + [out_data[1]]

    def CreateOperatorEx(self, ctx=None, in_shape=None, in_type=None):
        return DeformablePSROIPoolingOp(self.param_)


class DeformablePSROIPoolingOp(object):
    """DeformablePSROIPoolingOp::Forward (deformable_psroi_pooling-inl.h:64-95)."""

    def __init__(self, param):
        self.param_ = param

    def Forward(self, ctx, in_data, req, out_data, aux_args=None):
        p = self.param_
        if len(in_data) != (2 if p.no_trans else 3) or len(out_data) != 2:
            raise ValueError("DeformablePSROIPooling: wrong number of inputs / outputs")
        data, rois = in_data[0], in_data[1]
        if out_data[0].shape[0] != rois.shape[0] or out_data[1].shape[0] != rois.shape[0]:
            raise ValueError("DeformablePSROIPooling: output rows must equal the number of rois")
        for t in (data, rois, out_data[0], out_data[1]):
            if not t.is_contiguous():
                raise ValueError("DeformablePSROIPooling: tensors must be contiguous (deformable_psroi_pooling-inl.h:82-85)")
        trans = None if p.no_trans else in_data[2].float().contiguous()
        out, cnt = ops.deformable_psroi_pool(data, rois.float(), trans, p.spatial_scale, p.output_dim, p.group_size,
                                             p.pooled_size, p.part_size, p.sample_per_part, p.trans_std, p.no_trans,
                                             want_top_count=True)
        out_data[0].copy_(out)
        out_data[1].copy_(cnt)

    def Backward(self, ctx, out_grad, in_data, out_data, req, in_grad, aux_args=None):
        """DeformablePSROIPoolingOp::Backward (deformable_psroi_pooling-inl.h:97-140): in_grad = [d data, d rois(, d trans)];
        rois receive no gradient (the reference never writes grad_roi); kWriteInplace is rejected like there."""
        p = self.param_
        n_in = 2 if p.no_trans else 3
        if len(in_data) != n_in or len(out_data) != 2 or len(in_grad) != n_in:
            raise ValueError("DeformablePSROIPooling.Backward: wrong number of inputs / outputs")
        if req[0] == kWriteInplace or req[1] == kWriteInplace:
            raise ValueError("DeformablePSROIPooling: Backward doesn't support kWriteInplace.")
        data, rois = in_data[0], in_data[1]
        if out_grad[0].shape[0] != rois.shape[0] or out_data[1].shape[0] != rois.shape[0]:
            raise ValueError("DeformablePSROIPooling.Backward: rows of out_grad / top_count must equal the number of rois")
        trans = None if p.no_trans else in_data[2].float().contiguous()
        g = out_grad[0].to(data.dtype).contiguous()
        gdata, gtrans = ops.deformable_psroi_pool_bwd(g, data, rois.float().contiguous(), trans, p.spatial_scale, p.output_dim,
                                                      p.group_size, p.pooled_size, p.part_size, p.sample_per_part, p.trans_std, p.no_trans)
        _assign(in_grad[0], req[0], gdata)
        if not p.no_trans:
            _assign(in_grad[2], req[2], gtrans)


class contrib(object):
    """Eager counterparts of `mx.contrib.sym.DeformableConvolution` / `DeformablePSROIPooling`."""

    @staticmethod
    def DeformableConvolution(data, offset, weight, bias=None, name=None, **attrs):
        prop = DeformableConvolutionProp(**attrs)
        ins = [data, offset, weight] + ([] if prop.param_.no_bias else [bias])
        if not prop.param_.no_bias and bias is None:
            raise ValueError("DeformableConvolution: bias is required unless no_bias=True")
        _, (oshape,) = prop.InferShape([tuple(t.shape) for t in ins])
        if tuple(weight.shape) != (prop.param_.num_filter, data.shape[1], prop.param_.kernel[0], prop.param_.kernel[1]):
            raise ValueError("DeformableConvolution: weight shape %s" % (tuple(weight.shape),))
        out = torch.empty(oshape, device=data.device, dtype=data.dtype)
        prop.CreateOperatorEx().Forward(None, ins, [kWriteTo], [out])
        return out

    @staticmethod
    def DeformablePSROIPooling(data, rois, trans=None, name=None, **attrs):
        prop = DeformablePSROIPoolingProp(**attrs)
        ins = [data, rois] + ([] if prop.param_.no_trans else [trans])
        _, (oshape, _) = prop.InferShape([tuple(t.shape) for t in ins])
        out = torch.empty(oshape, device=data.device, dtype=data.dtype)
        cnt = torch.empty(oshape, device=data.device, dtype=data.dtype)
        prop.CreateOperatorEx().Forward(None, ins, [kWriteTo, kWriteTo], [out, cnt])
        return out                                  # one visible output (NumVisibleOutputs = 1)
