// Test harness for include/relnet_operator_cxx.hpp: plain-C entry points (called from pytest through ctypes) that drive
// the OperatorProperty / Operator classes exactly as an MXNet operator body would -- Prop::InferShape, CreateOperatorEx,
// Op::Forward / Op::Backward on raw device pointers, kTempSpace served from hipMalloc.
#include <cstdio>
#include <cstring>
#include <memory>
#include <vector>

#include "relnet_operator_cxx.hpp"

using namespace relnet_op;

namespace {
struct Arena {
  std::vector<void*> blocks;
  void* get(size_t bytes) { void* p = nullptr; if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) throw std::runtime_error("hipMalloc"); blocks.push_back(p); return p; }
  ~Arena() { (void)hipDeviceSynchronize(); for (void* p : blocks) (void)hipFree(p); }
};
char g_err[1024] = "";
}  // namespace

extern "C" const char* opcxx_last_error() { return g_err; }

// data [N,C,H,W], offset [N,2*k*k*dg,Ho,Wo], weight [Co,C,k,k], bias [Co] or null, out [N,Co,Ho,Wo]
// backward: dy -> gdata, goffset, gweight, gbias (req codes: 0 null, 1 write, 3 add)
extern "C" int opcxx_deformable_conv(const float* data, const float* offset, const float* weight, const float* bias, float* out,
                                     const float* dy, float* gdata, float* goffset, float* gweight, float* gbias, const int* req4,
                                     int N, int C, int H, int W, int Co, int k, int pad, int stride, int dil, int dg, long* out_shape4) {
  try {
    DeformableConvolutionParam p;
    p.kernel = {k, k}; p.stride = {stride, stride}; p.dilate = {dil, dil}; p.pad = {pad, pad};
    p.num_filter = (uint32_t)Co; p.num_deformable_group = (uint32_t)dg; p.no_bias = bias == nullptr;
    DeformableConvolutionProp prop(p);
    if (prop.TypeString() != "_contrib_DeformableConvolution" || prop.ListArguments().size() != (bias ? 4u : 3u)) return -2;
    const long Ho = (H + 2 * pad - (dil * (k - 1) + 1)) / stride + 1, Wo = (W + 2 * pad - (dil * (k - 1) + 1)) / stride + 1;
    std::vector<TShape> in = {{N, C, H, W}, {N, 2L * k * k * dg, Ho, Wo}, {}};
    if (bias) in.push_back({});
    std::vector<TShape> outs;
    prop.InferShape(&in, &outs);
    for (int i = 0; i < 4; ++i) out_shape4[i] = outs[0][i];
    if (in[2] != TShape({Co, C, k, k})) return -3;
    std::unique_ptr<DeformableConvolutionOp> op(prop.CreateOperatorEx());
    Arena arena;
    OpContext ctx;
    ctx.stream = nullptr;
    ctx.temp_space = [&](size_t b) { return arena.get(b); };
    std::vector<TBlob> in_data = {TBlob((void*)data, in[0]), TBlob((void*)offset, in[1]), TBlob((void*)weight, in[2])};
    if (bias) in_data.push_back(TBlob((void*)bias, in[3]));
    std::vector<TBlob> out_data = {TBlob(out, outs[0])};
    op->Forward(ctx, in_data, {kWriteTo}, out_data);
    if (dy) {
      std::vector<TBlob> in_grad = {TBlob(gdata, in[0]), TBlob(goffset, in[1]), TBlob(gweight, in[2])};
      std::vector<OpReqType> req = {(OpReqType)req4[0], (OpReqType)req4[1], (OpReqType)req4[2]};
      if (bias) { in_grad.push_back(TBlob(gbias, in[3])); req.push_back((OpReqType)req4[3]); }
      op->Backward(ctx, {TBlob((void*)dy, outs[0])}, in_data, out_data, req, in_grad);
    }
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    return 0;
  } catch (const std::exception& e) {
    std::snprintf(g_err, sizeof(g_err), "%s", e.what());
    return -1;
  }
}

extern "C" int opcxx_deformable_psroi(const float* data, const float* rois, const float* trans, float* out, float* top_count, const float* dy,
                                      float* gdata, float* gtrans, int N, int C, int H, int W, int R, int output_dim, int group, int pooled,
                                      int spp, float scale, float trans_std, int ncls) {
  try {
    DeformablePSROIPoolingParam p;
    p.spatial_scale = scale; p.output_dim = output_dim; p.group_size = group; p.pooled_size = pooled; p.part_size = 0;
    p.sample_per_part = spp; p.trans_std = trans_std; p.no_trans = trans == nullptr;
    DeformablePSROIPoolingProp prop(p);
    std::vector<TShape> in = {{N, C, H, W}, {R, 5}};
    if (trans) in.push_back({R, 2L * ncls, pooled, pooled});
    std::vector<TShape> outs;
    prop.InferShape(&in, &outs);
    if (outs.size() != 2 || outs[0] != TShape({R, output_dim, pooled, pooled}) || prop.NumVisibleOutputs() != 1) return -3;
    std::unique_ptr<DeformablePSROIPoolingOp> op(prop.CreateOperatorEx());
    OpContext ctx;
    std::vector<TBlob> in_data = {TBlob((void*)data, in[0]), TBlob((void*)rois, in[1])};
    if (trans) in_data.push_back(TBlob((void*)trans, in[2]));
    std::vector<TBlob> out_data = {TBlob(out, outs[0]), TBlob(top_count, outs[1])};
    op->Forward(ctx, in_data, {kWriteTo, kWriteTo}, out_data);
    if (dy) {
      std::vector<TBlob> in_grad = {TBlob(gdata, in[0]), TBlob(nullptr, in[1])};
      std::vector<OpReqType> req = {kWriteTo, kNullOp};
      if (trans) { in_grad.push_back(TBlob(gtrans, in[2])); req.push_back(kWriteTo); }
      op->Backward(ctx, {TBlob((void*)dy, outs[0])}, in_data, out_data, req, in_grad);
    }
    if (hipDeviceSynchronize() != hipSuccess) return -4;
    return 0;
  } catch (const std::exception& e) {
    std::snprintf(g_err, sizeof(g_err), "%s", e.what());
    return -1;
  }
}
