// C++ binding of the reference's `operator_cxx` interface over the C-ABI of librelnet_hip.so.
//
// The reference registers two legacy MXNet operators for its DCN configuration
//   _contrib_DeformableConvolution     relation_rcnn/operator_cxx/deformable_convolution-inl.h:39-470
//   _contrib_DeformablePSROIPooling    relation_rcnn/operator_cxx/deformable_psroi_pooling-inl.h:32-270
// as `XParam` (dmlc parameter struct) + `XProp : OperatorProperty` + `XOp : Operator`.  This header restates that
// shape -- the same class and method names, argument order, blob indices (conv::kData ...), OpReqType semantics and
// CHECK conditions -- without MXNet / mshadow / dmlc, so that an MXNet build can forward its operator bodies to
// these classes one to one (TBlob::dptr_ -> relnet_op::TBlob::dptr_, ctx.get_stream<gpu>() -> OpContext::stream,
// ctx.requested[kTempSpace] -> OpContext::temp_space) and a plain C++ host can call them directly.
// Blobs are float32 NCHW device memory, as in the reference.  Failed CHECKs throw std::runtime_error (MXNet aborts).
#ifndef RELNET_OPERATOR_CXX_HPP
#define RELNET_OPERATOR_CXX_HPP

#include <cstddef>
#include <cstdint>
#include <functional>
#include <stdexcept>
#include <string>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "relnet_hip.h"

namespace relnet_op {

enum OpReqType { kNullOp = 0, kWriteTo = 1, kWriteInplace = 2, kAddTo = 3 };
enum { kFloat32 = 0 };

typedef std::vector<long> TShape;
inline long ShapeSize(const TShape& s) { long n = 1; for (long d : s) n *= d; return n; }

struct TBlob {
  void* dptr_ = nullptr;
  TShape shape_;
  int type_flag_ = kFloat32;
  TBlob() {}
  TBlob(void* p, const TShape& s) : dptr_(p), shape_(s) {}
  long Size() const { return ShapeSize(shape_); }
};

// OpContext: the stream the kernels are enqueued on and the kTempSpace resource (a device allocation that lives
// until the stream has drained; the reference asks MXNet for it through ForwardResource / BackwardResource)
struct OpContext {
  void* stream = nullptr;
  std::function<void*(size_t)> temp_space;
};

#define RELNET_OP_CHECK(cond, msg) do { if (!(cond)) throw std::runtime_error(std::string(msg)); } while (0)
inline void relnet_call(int rc) { if (rc != 0) throw std::runtime_error(std::string("librelnet_hip: ") + relnet_last_error()); }
inline void relnet_hipcheck(hipError_t e) { if (e != hipSuccess) throw std::runtime_error(hipGetErrorString(e)); }

namespace conv { enum { kData = 0, kOffset = 1, kWeight = 2, kBias = 3, kOut = 0 }; }
namespace deformablepsroipool { enum { kData = 0, kBox = 1, kTrans = 2, kOut = 0, kTopCount = 1 }; }

// ------------------------------------------------------------------------------------------------------------------
struct DeformableConvolutionParam {             // deformable_convolution-inl.h:39-76
  TShape kernel, stride, dilate, pad;
  uint32_t num_filter = 0, num_group = 1, num_deformable_group = 1;
  uint64_t workspace = 1024;
  bool no_bias = false;
};

class DeformableConvolutionOp {
 public:
  explicit DeformableConvolutionOp(const DeformableConvolutionParam& p) : param_(p) {
    RELNET_OP_CHECK(p.num_group == 1, "DeformableConvolution: num_group > 1 is not built (unused by the reference graphs)");
  }

  // deformable_convolution-inl.h:91-143
  void Forward(const OpContext& ctx, const std::vector<TBlob>& in_data, const std::vector<OpReqType>& req,
               const std::vector<TBlob>& out_data, const std::vector<TBlob>& aux_args = {}) {
    RELNET_OP_CHECK(req[conv::kOut] == kWriteTo, "DeformableConvolution: req[kOut] must be kWriteTo");
    const size_t expected = param_.no_bias ? 3 : 4;
    RELNET_OP_CHECK(in_data.size() == expected && out_data.size() == 1, "DeformableConvolution: wrong number of blobs");
    Setup(in_data[conv::kData].shape_, in_data[conv::kOffset].shape_, out_data[conv::kOut].shape_);
    float* col = Temp<float>(ctx, (size_t)P_ * K_);
    float* wpack = Temp<float>(ctx, (size_t)Co_ * K_);
    float* out_nhwc = Temp<float>(ctx, (size_t)P_ * Co_);
    PackWeight(ctx, (const float*)in_data[conv::kWeight].dptr_, wpack);
    Im2col(ctx, in_data, col);
    const float* bias = param_.no_bias ? nullptr : (const float*)in_data[conv::kBias].dptr_;
    relnet_call(relnet_gemm_nt(col, K_, 0, wpack, K_, 0, out_nhwc, Co_, 0, bias, bias ? 1 : 0, nullptr, 0, (int)P_, (int)Co_,
                               (int)K_, 1, 0, 0, ctx.stream));
    // [N][Ho*Wo][Cout] -> NCHW
    relnet_call(relnet_transpose_2d(out_nhwc, Co_, HoWo_ * Co_, out_data[conv::kOut].dptr_, HoWo_, Co_ * HoWo_, (int)HoWo_, (int)Co_,
                                    (int)N_, 0, ctx.stream));
  }

  // deformable_convolution-inl.h:145-237
  void Backward(const OpContext& ctx, const std::vector<TBlob>& out_grad, const std::vector<TBlob>& in_data,
                const std::vector<TBlob>& out_data, const std::vector<OpReqType>& req, const std::vector<TBlob>& in_grad,
                const std::vector<TBlob>& aux_args = {}) {
    RELNET_OP_CHECK(out_grad.size() == 1, "DeformableConvolution.Backward: one out_grad");
    const size_t expected = param_.no_bias ? 3 : 4;
    RELNET_OP_CHECK(in_data.size() == expected && in_grad.size() == expected && req.size() == expected,
                    "DeformableConvolution.Backward: wrong number of blobs");
    Setup(in_grad[conv::kData].shape_, in_grad[conv::kOffset].shape_, out_grad[conv::kOut].shape_);
    hipStream_t s = (hipStream_t)ctx.stream;
    const long Ppad = (P_ + 15) / 16 * 16;
    float* dy_nhwc = Temp<float>(ctx, (size_t)P_ * Co_);
    float* wpack = Temp<float>(ctx, (size_t)Co_ * K_);
    float* wpack_t = Temp<float>(ctx, (size_t)K_ * Co_);
    float* dcol = Temp<float>(ctx, (size_t)P_ * K_);
    // NCHW gradient -> [P][Cout]
    relnet_call(relnet_transpose_2d(out_grad[conv::kOut].dptr_, HoWo_, Co_ * HoWo_, dy_nhwc, Co_, HoWo_ * Co_, (int)Co_, (int)HoWo_,
                                    (int)N_, 0, s));
    PackWeight(ctx, (const float*)in_data[conv::kWeight].dptr_, wpack);
    relnet_call(relnet_transpose_2d(wpack, K_, 0, wpack_t, Co_, 0, (int)Co_, (int)K_, 1, 0, s));
    // col_buffer = W^T . out_grad  (:196-198), as rows: dcol [P][K] = dY [P][Cout] . W [Cout][K]
    relnet_call(relnet_gemm_nt(dy_nhwc, Co_, 0, wpack_t, Co_, 0, dcol, K_, 0, nullptr, 0, nullptr, 0, (int)P_, (int)K_, (int)Co_, 1, 0, 0, s));
    // gradients w.r.t. the sampling offsets and the input (deformable_col2im_coord / deformable_col2im, :201-215)
    if (req[conv::kData] != kNullOp || req[conv::kOffset] != kNullOp) {
      float* gdata = (float*)in_grad[conv::kData].dptr_;
      float* goff = (float*)in_grad[conv::kOffset].dptr_;
      float* scratch_d = nullptr;
      float* scratch_o = nullptr;
      if (req[conv::kData] == kNullOp) gdata = scratch_d = Temp<float>(ctx, (size_t)N_ * C_ * H_ * W_);
      if (req[conv::kOffset] == kNullOp) goff = scratch_o = Temp<float>(ctx, (size_t)ShapeSize(in_grad[conv::kOffset].shape_));
      if (req[conv::kData] != kAddTo) relnet_hipcheck(hipMemsetAsync(gdata, 0, sizeof(float) * N_ * C_ * H_ * W_, s));     // data_grad = 0 (:189-190)
      if (req[conv::kOffset] != kAddTo) relnet_hipcheck(hipMemsetAsync(goff, 0, sizeof(float) * ShapeSize(in_grad[conv::kOffset].shape_), s));
      const long ds[4] = {C_ * H_ * W_, H_ * W_, W_, 1};
      const long os[4] = {OffC_ * HoWo_, HoWo_, Wo_, 1};
      relnet_call(relnet_deformable_col2im(dcol, K_, 0, in_data[conv::kData].dptr_, ds, 0, (const float*)in_data[conv::kOffset].dptr_, os,
                                           gdata, ds, goff, os, (int)N_, (int)C_, (int)H_, (int)W_, (int)param_.kernel[0], (int)param_.kernel[1],
                                           (int)param_.pad[0], (int)param_.pad[1], (int)param_.stride[0], (int)param_.stride[1],
                                           (int)param_.dilate[0], (int)param_.dilate[1], (int)param_.num_deformable_group, s));
    }
    // transposed gradient, zero padded along the contraction (pixels): shared by dWeight and dBias
    float* dy_t = nullptr;
    if (req[conv::kWeight] != kNullOp || (!param_.no_bias && req[conv::kBias] != kNullOp)) {
      dy_t = Temp<float>(ctx, (size_t)Co_ * Ppad);
      relnet_hipcheck(hipMemsetAsync(dy_t, 0, sizeof(float) * Co_ * Ppad, s));
      relnet_call(relnet_transpose_2d(dy_nhwc, Co_, 0, dy_t, Ppad, 0, (int)P_, (int)Co_, 1, 0, s));
    }
    // gradient w.r.t. the weight: dW [Cout][K] = dY^T . col over all pixels of the batch (:217-226)
    if (req[conv::kWeight] != kNullOp) {
      float* col = dcol;                                   // the column gradient is consumed: reuse its buffer
      Im2col(ctx, in_data, col);
      float* col_t = Temp<float>(ctx, (size_t)K_ * Ppad);
      relnet_hipcheck(hipMemsetAsync(col_t, 0, sizeof(float) * K_ * Ppad, s));
      relnet_call(relnet_transpose_2d(col, K_, 0, col_t, Ppad, 0, (int)P_, (int)K_, 1, 0, s));
      float* dw_pack = wpack;                              // [Cout][kh*kw][C], reuses the packed-weight buffer
      const float* resid = nullptr;
      if (req[conv::kWeight] == kAddTo) {                  // accumulate onto the caller's gradient: it rides in the GEMM epilogue
        PackWeight(ctx, (const float*)in_grad[conv::kWeight].dptr_, wpack_t);
        resid = wpack_t;
      }
      relnet_call(relnet_gemm_nt(dy_t, Ppad, 0, col_t, Ppad, 0, dw_pack, K_, 0, nullptr, 0, resid, 0, (int)Co_, (int)K_, (int)Ppad, 1, 0, 0, s));
      const long kk = param_.kernel[0] * param_.kernel[1];
      relnet_call(relnet_transpose_2d(dw_pack, C_, K_, in_grad[conv::kWeight].dptr_, kk, K_, (int)kk, (int)C_, (int)Co_, 0, s));
    }
    // gradient w.r.t. the bias: sumall_except_dim<1> (:228-233) = dY^T . 1
    if (!param_.no_bias && req[conv::kBias] != kNullOp) {
      float* ones = Temp<float>(ctx, (size_t)Ppad);
      std::vector<float> h((size_t)Ppad, 0.f);
      for (long i = 0; i < P_; ++i) h[(size_t)i] = 1.f;
      relnet_hipcheck(hipMemcpyAsync(ones, h.data(), sizeof(float) * Ppad, hipMemcpyHostToDevice, s));
      relnet_hipcheck(hipStreamSynchronize(s));            // `h` leaves scope
      const float* resid = req[conv::kBias] == kAddTo ? (const float*)in_grad[conv::kBias].dptr_ : nullptr;
      relnet_call(relnet_gemm_nt(dy_t, Ppad, 0, ones, Ppad, 0, in_grad[conv::kBias].dptr_, 1, 0, nullptr, 0, resid, 0, (int)Co_, 1, (int)Ppad,
                                 1, 0, 0, s));
    }
  }

 private:
  template <typename T> T* Temp(const OpContext& ctx, size_t n) {
    RELNET_OP_CHECK((bool)ctx.temp_space, "OpContext::temp_space (kTempSpace) is required");
    return (T*)ctx.temp_space(n * sizeof(T));
  }
  void Setup(const TShape& ishape, const TShape& offset_shape, const TShape& oshape) {          // LayerSetUp (:240-267)
    RELNET_OP_CHECK(ishape.size() == 4 && offset_shape.size() == 4 && oshape.size() == 4, "4-D NCHW blobs expected");
    RELNET_OP_CHECK(param_.kernel.size() == 2, "only 2-D deformable convolution is supported");
    N_ = ishape[0]; C_ = ishape[1]; H_ = ishape[2]; W_ = ishape[3];
    Co_ = oshape[1]; Ho_ = oshape[2]; Wo_ = oshape[3];
    HoWo_ = Ho_ * Wo_; P_ = N_ * HoWo_; K_ = param_.kernel[0] * param_.kernel[1] * C_;
    OffC_ = offset_shape[1];
    RELNET_OP_CHECK(K_ % 16 == 0 && Co_ % 16 == 0, "float32 MFMA GEMM granularity: C and num_filter must be multiples of 16");
  }
  void PackWeight(const OpContext& ctx, const float* w_oihw, float* packed) {                   // [O][C][kh*kw] -> [O][kh*kw][C]
    const long kk = param_.kernel[0] * param_.kernel[1];
    relnet_call(relnet_transpose_2d(w_oihw, kk, K_, packed, C_, K_, (int)C_, (int)kk, (int)Co_, 0, ctx.stream));
  }
  void Im2col(const OpContext& ctx, const std::vector<TBlob>& in_data, float* col) {
    const long ds[4] = {C_ * H_ * W_, H_ * W_, W_, 1};
    const long os[4] = {OffC_ * HoWo_, HoWo_, Wo_, 1};
    relnet_call(relnet_deformable_im2col(in_data[conv::kData].dptr_, ds, (const float*)in_data[conv::kOffset].dptr_, os, col, K_, (int)N_,
                                         (int)C_, (int)H_, (int)W_, (int)param_.kernel[0], (int)param_.kernel[1], (int)param_.pad[0],
                                         (int)param_.pad[1], (int)param_.stride[0], (int)param_.stride[1], (int)param_.dilate[0],
                                         (int)param_.dilate[1], (int)param_.num_deformable_group, 0, 0, ctx.stream));
  }
  DeformableConvolutionParam param_;
  long N_ = 0, C_ = 0, H_ = 0, W_ = 0, Co_ = 0, Ho_ = 0, Wo_ = 0, HoWo_ = 0, P_ = 0, K_ = 0, OffC_ = 0;
};

class DeformableConvolutionProp {                  // deformable_convolution-inl.h:294-470
 public:
  explicit DeformableConvolutionProp(const DeformableConvolutionParam& p) : param_(p) {
    if (param_.stride.empty()) param_.stride = {1, 1};
    if (param_.dilate.empty()) param_.dilate = {1, 1};
    if (param_.pad.empty()) param_.pad = {0, 0};
  }
  std::vector<std::string> ListArguments() const {
    if (!param_.no_bias) return {"data", "offset", "weight", "bias"};
    return {"data", "offset", "weight"};
  }
  std::vector<std::string> ListOutputs() const { return {"output"}; }
  std::string TypeString() const { return "_contrib_DeformableConvolution"; }

  bool InferShape(std::vector<TShape>* in_shape, std::vector<TShape>* out_shape, std::vector<TShape>* aux_shape = nullptr) const {
    const size_t expected = param_.no_bias ? 3 : 4;
    RELNET_OP_CHECK(in_shape->size() == expected, param_.no_bias ? "Input:[data, offset, weight]" : "Input:[data, offset, weight, bias]");
    out_shape->resize(1, TShape());
    const TShape& d = (*in_shape)[conv::kData];
    const TShape& o = (*in_shape)[conv::kOffset];
    if (d.empty()) return false;
    RELNET_OP_CHECK(param_.kernel.size() == 2, "not implemented");                                  // :413-416
    RELNET_OP_CHECK(d.size() == 4, "Input data should be 4D in batch-num_filter-y-x");
    RELNET_OP_CHECK(o.size() == 4, "Input offset should be 4D in batch-num_filter-y-x");
    RELNET_OP_CHECK(d[1] % param_.num_group == 0, "input num_filter must divide group size");
    RELNET_OP_CHECK(d[1] % param_.num_deformable_group == 0, "input num_filter must divide deformable group size");
    RELNET_OP_CHECK(param_.num_filter % param_.num_group == 0, "output num_filter must divide group size");
    const long ky = param_.kernel[0], kx = param_.kernel[1];
    RELNET_OP_CHECK(ky * kx > 0 && param_.stride[0] * param_.stride[1] > 0 && param_.dilate[0] * param_.dilate[1] > 0,
                    "incorrect kernel / stride / dilate size");
    (*in_shape)[conv::kWeight] = {(long)param_.num_filter, d[1] / (long)param_.num_group, ky, kx};
    if (!param_.no_bias) (*in_shape)[conv::kBias] = {(long)param_.num_filter};
    TShape out = {d[0], (long)param_.num_filter, (d[2] + 2 * param_.pad[0] - (param_.dilate[0] * (ky - 1) + 1)) / param_.stride[0] + 1,
                  (d[3] + 2 * param_.pad[1] - (param_.dilate[1] * (kx - 1) + 1)) / param_.stride[1] + 1};
    RELNET_OP_CHECK(out[1] % param_.num_deformable_group == 0, "output num_filter must divide deformable group size");
    RELNET_OP_CHECK(out[2] == o[2], "output height must equal to offset map height");
    RELNET_OP_CHECK(out[3] == o[3], "output width must equal to offset map width");
    RELNET_OP_CHECK(o[1] % (ky * kx) == 0, "offset filter must divide deformable group size");
    RELNET_OP_CHECK(o[1] / (2 * ky * kx) == (long)param_.num_deformable_group, "offset filter must divide deformable group size");
    RELNET_OP_CHECK(ky <= d[2] + 2 * param_.pad[0] && kx <= d[3] + 2 * param_.pad[1], "kernel size exceed input");
    (*out_shape)[0] = out;
    return true;
  }
  bool InferType(std::vector<int>* in_type, std::vector<int>* out_type, std::vector<int>* aux_type = nullptr) const {
    RELNET_OP_CHECK(!in_type->empty() && (*in_type)[0] != -1, "First input must have specified type");
    const int dtype = (*in_type)[0];
    for (size_t i = 0; i < in_type->size(); ++i) {
      if ((*in_type)[i] == -1) (*in_type)[i] = dtype;
      else RELNET_OP_CHECK((*in_type)[i] == dtype, "This layer requires uniform type");
    }
    out_type->assign(1, dtype);
    return true;
  }
  std::vector<int> DeclareBackwardDependency(const std::vector<int>& out_grad, const std::vector<int>& in_data,
                                             const std::vector<int>& out_data) const {
    return {out_grad[conv::kOut], in_data[conv::kData], in_data[conv::kOffset], in_data[conv::kWeight]};
  }
  // bytes of kTempSpace one Forward / Backward call asks for (ForwardResource / BackwardResource, :455-463)
  size_t ForwardResource(const std::vector<TShape>& in_shape) const { return Workspace(in_shape, false); }
  size_t BackwardResource(const std::vector<TShape>& in_shape) const { return Workspace(in_shape, true); }
  DeformableConvolutionOp* CreateOperatorEx() const { return new DeformableConvolutionOp(param_); }

 private:
  size_t Workspace(const std::vector<TShape>& in_shape, bool bwd) const {
    std::vector<TShape> in = in_shape, out;
    InferShape(&in, &out);
    const long P = out[0][0] * out[0][2] * out[0][3], K = param_.kernel[0] * param_.kernel[1] * in[0][1], Co = out[0][1];
    const long Ppad = (P + 15) / 16 * 16;
    size_t n = (size_t)P * K + (size_t)Co * K + (size_t)P * Co;
    if (bwd) n += (size_t)K * Co + (size_t)Co * Ppad + (size_t)K * Ppad + Ppad + (size_t)ShapeSize(in[0]) + (size_t)ShapeSize(in[1]);
    return n * sizeof(float) + 16 * 256;
  }
  DeformableConvolutionParam param_;
};

// ------------------------------------------------------------------------------------------------------------------
struct DeformablePSROIPoolingParam {            // deformable_psroi_pooling-inl.h:32-55
  float spatial_scale = 0.f;
  int output_dim = 0, group_size = 0, pooled_size = 0, part_size = 0, sample_per_part = 1;
  float trans_std = 0.f;
  bool no_trans = false;
};

class DeformablePSROIPoolingOp {
 public:
  explicit DeformablePSROIPoolingOp(const DeformablePSROIPoolingParam& p) : param_(p) {}

  // deformable_psroi_pooling-inl.h:64-95
  void Forward(const OpContext& ctx, const std::vector<TBlob>& in_data, const std::vector<OpReqType>& req,
               const std::vector<TBlob>& out_data, const std::vector<TBlob>& aux_args = {}) {
    const size_t in_expected = param_.no_trans ? 2 : 3;
    RELNET_OP_CHECK(in_data.size() == in_expected && out_data.size() == 2, "DeformablePSROIPooling: wrong number of blobs");
    const TShape& d = in_data[deformablepsroipool::kData].shape_;
    const TShape& o = out_data[deformablepsroipool::kOut].shape_;
    RELNET_OP_CHECK(o[0] == in_data[deformablepsroipool::kBox].shape_[0] &&
                    out_data[deformablepsroipool::kTopCount].shape_[0] == o[0], "output rows must equal the number of rois");
    const long ds[4] = {d[1] * d[2] * d[3], d[2] * d[3], d[3], 1};
    const long os[4] = {o[1] * o[2] * o[3], o[2] * o[3], o[3], 1};
    const float* trans = param_.no_trans ? nullptr : (const float*)in_data[deformablepsroipool::kTrans].dptr_;
    const int ncls = param_.no_trans ? 0 : (int)(in_data[deformablepsroipool::kTrans].shape_[1] / 2);
    relnet_call(relnet_deformable_psroi_pool_fwd(in_data[deformablepsroipool::kData].dptr_, ds, (const float*)in_data[deformablepsroipool::kBox].dptr_,
                                                 trans, out_data[deformablepsroipool::kOut].dptr_, os,
                                                 (float*)out_data[deformablepsroipool::kTopCount].dptr_, (int)o[0], (int)d[1], (int)d[2], (int)d[3],
                                                 param_.output_dim, param_.group_size, param_.pooled_size, param_.part_size, param_.sample_per_part,
                                                 param_.spatial_scale, param_.trans_std, ncls, 0, 0, ctx.stream));
  }

  // deformable_psroi_pooling-inl.h:97-140
  void Backward(const OpContext& ctx, const std::vector<TBlob>& out_grad, const std::vector<TBlob>& in_data,
                const std::vector<TBlob>& out_data, const std::vector<OpReqType>& req, const std::vector<TBlob>& in_grad,
                const std::vector<TBlob>& aux_args = {}) {
    const size_t in_expected = param_.no_trans ? 2 : 3;
    RELNET_OP_CHECK(in_data.size() == in_expected && out_data.size() == 2, "DeformablePSROIPooling.Backward: wrong number of blobs");
    RELNET_OP_CHECK(out_grad[deformablepsroipool::kOut].shape_[0] == in_data[deformablepsroipool::kBox].shape_[0], "rows of out_grad != rois");
    RELNET_OP_CHECK(req[deformablepsroipool::kData] != kWriteInplace && req[deformablepsroipool::kBox] != kWriteInplace,
                    "DeformablePSROIPooling: Backward doesn't support kWriteInplace.");
    hipStream_t s = (hipStream_t)ctx.stream;
    const TShape& d = in_data[deformablepsroipool::kData].shape_;
    const TShape& o = out_grad[deformablepsroipool::kOut].shape_;
    const long ds[4] = {d[1] * d[2] * d[3], d[2] * d[3], d[3], 1};
    const long os[4] = {o[1] * o[2] * o[3], o[2] * o[3], o[3], 1};
    float* gdata = (float*)in_grad[deformablepsroipool::kData].dptr_;
    if (req[deformablepsroipool::kData] != kAddTo)                                                  // Assign(grad_in, req, 0), :135
      relnet_hipcheck(hipMemsetAsync(gdata, 0, sizeof(float) * ShapeSize(d), s));
    const float* trans = nullptr;
    float* gtrans = nullptr;
    int ncls = 0;
    if (!param_.no_trans) {
      RELNET_OP_CHECK(in_grad.size() == 3, "DeformablePSROIPooling.Backward: in_grad needs the trans gradient");
      trans = (const float*)in_data[deformablepsroipool::kTrans].dptr_;
      gtrans = (float*)in_grad[deformablepsroipool::kTrans].dptr_;
      ncls = (int)(in_data[deformablepsroipool::kTrans].shape_[1] / 2);
      if (req[deformablepsroipool::kTrans] != kAddTo)
        relnet_hipcheck(hipMemsetAsync(gtrans, 0, sizeof(float) * ShapeSize(in_data[deformablepsroipool::kTrans].shape_), s));
    }
    relnet_call(relnet_deformable_psroi_pool_bwd(out_grad[deformablepsroipool::kOut].dptr_, os, in_data[deformablepsroipool::kData].dptr_, ds,
                                                 (const float*)in_data[deformablepsroipool::kBox].dptr_, trans, gdata, ds, gtrans, (int)o[0], (int)d[1],
                                                 (int)d[2], (int)d[3], param_.output_dim, param_.group_size, param_.pooled_size, param_.part_size,
                                                 param_.sample_per_part, param_.spatial_scale, param_.trans_std, ncls, 0, 0, s));
  }

 private:
  DeformablePSROIPoolingParam param_;
};

class DeformablePSROIPoolingProp {                 // deformable_psroi_pooling-inl.h:153-270
 public:
  explicit DeformablePSROIPoolingProp(const DeformablePSROIPoolingParam& p) : param_(p) {
    if (param_.part_size == 0) param_.part_size = param_.pooled_size;                                 // Init(), :176-180
  }
  std::vector<std::string> ListArguments() const {
    if (param_.no_trans) return {"data", "rois"};
    return {"data", "rois", "trans"};
  }
  std::vector<std::string> ListOutputs() const { return {"output", "top_count"}; }
  int NumOutputs() const { return 2; }
  int NumVisibleOutputs() const { return 1; }
  std::string TypeString() const { return "_contrib_DeformablePSROIPooling"; }
  bool InferShape(std::vector<TShape>* in_shape, std::vector<TShape>* out_shape, std::vector<TShape>* aux_shape = nullptr) const {
    RELNET_OP_CHECK(in_shape->size() == (param_.no_trans ? 2u : 3u), param_.no_trans ? "Input:[data, rois]" : "Input:[data, rois, trans]");
    const TShape& d = (*in_shape)[deformablepsroipool::kData];
    const TShape& b = (*in_shape)[deformablepsroipool::kBox];
    RELNET_OP_CHECK(d.size() == 4, "data should be a 4D tensor");
    RELNET_OP_CHECK(b.size() == 2 && b[1] == 5, "bbox should be a 2D tensor of shape [batch, 5]");
    const TShape o = {b[0], (long)param_.output_dim, (long)param_.pooled_size, (long)param_.pooled_size};
    out_shape->assign(2, o);
    return true;
  }
  std::vector<int> DeclareBackwardDependency(const std::vector<int>& out_grad, const std::vector<int>& in_data,
                                             const std::vector<int>& out_data) const {
    if (param_.no_trans) return {out_grad[0], in_data[0], in_data[1], out_data[1]};
    return {out_grad[0], in_data[0], in_data[1], in_data[2], out_data[1]};
  }
  DeformablePSROIPoolingOp* CreateOperatorEx() const { return new DeformablePSROIPoolingOp(param_); }

 private:
  DeformablePSROIPoolingParam param_;
};

}  // namespace relnet_op
#endif  // RELNET_OPERATOR_CXX_HPP
