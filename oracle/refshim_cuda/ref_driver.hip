// oracle/_ref driver (TEST INFRASTRUCTURE ONLY): C entry points around the reference's own CUDA kernels, which are
// #included below from /root/reference UNEDITED (hipcc compiles the `__global__` bodies for gfx950; refshim.h supplies the
// few MXNet / mshadow / CUDA-runtime names they mention).  Launch geometry is the reference's: cuda_get_num_blocks(n)
// blocks of mshadow::cuda::kBaseThreadNum threads (deformable_im2col.cuh:283-291,375-382,476-483;
// deformable_psroi_pooling.cu:170-176,333-341).  All pointers are HOST pointers; tensors are dense fp32, NCHW.
#include "refshim.h"
#include "nn/deformable_im2col.cuh"            // relation_rcnn/operator_cxx/nn/deformable_im2col.cuh
#include "deformable_psroi_pooling.cu"         // relation_rcnn/operator_cxx/deformable_psroi_pooling.cu
#include "gpu_nms.hpp"                         // lib/nms/gpu_nms.hpp: void _nms(...), defined by lib/nms/nms_kernel.cu (second TU)

namespace {
struct DevBuf {
  void* p = nullptr;
  DevBuf(const void* host, size_t bytes, bool zero = false) {
    if (hipMalloc(&p, bytes ? bytes : 4) != hipSuccess) { p = nullptr; return; }
    if (host) (void)hipMemcpy(p, host, bytes, hipMemcpyHostToDevice);
    else if (zero) (void)hipMemset(p, 0, bytes);
  }
  ~DevBuf() { if (p) (void)hipFree(p); }
  template <typename T> T* as() { return (T*)p; }
};
int finish(void* host, DevBuf& d, size_t bytes) {
  if (hipDeviceSynchronize() != hipSuccess || hipGetLastError() != hipSuccess) return -1;
  return hipMemcpy(host, d.p, bytes, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
int out_dim(int in, int k, int pad, int stride, int dil) { return (in + 2 * pad - (dil * (k - 1) + 1)) / stride + 1; }
using mxnet::op::mxnet_op::cuda_get_num_blocks;
const int kThreads = mshadow::cuda::kBaseThreadNum;
}  // namespace

extern "C" {

// data_im [C,H,W], offset [dg*2*kh*kw, Ho, Wo] -> col [C*kh*kw, Ho, Wo]        (one image, as DeformableConvolutionOp::Forward calls it)
int ref_deformable_im2col(const float* im, const float* offset, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw,
                          int dh, int dw, int dg, float* col) {
  const int Ho = out_dim(H, kh, ph, sh, dh), Wo = out_dim(W, kw, pw, sw, dw);
  const size_t ncol = (size_t)C * kh * kw * Ho * Wo;
  DevBuf dim(im, (size_t)C * H * W * 4), doff(offset, (size_t)dg * 2 * kh * kw * Ho * Wo * 4), dcol(nullptr, ncol * 4, true);
  const int n = C * Ho * Wo;
  mxnet::op::deformable_im2col_gpu_kernel<float><<<cuda_get_num_blocks(n), kThreads>>>(
      n, dim.as<float>(), doff.as<float>(), H, W, kh, kw, ph, pw, sh, sw, dh, dw, C / dg, Ho, Wo, dcol.as<float>());
  return finish(col, dcol, ncol * 4);
}

// col [C*kh*kw, Ho, Wo], offset -> grad_im [C,H,W] (accumulated onto zeros with the reference's atomicAdd)
int ref_deformable_col2im(const float* col, const float* offset, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw,
                          int dh, int dw, int dg, float* grad_im) {
  const int Ho = out_dim(H, kh, ph, sh, dh), Wo = out_dim(W, kw, pw, sw, dw);
  const size_t ncol = (size_t)C * kh * kw * Ho * Wo;
  DevBuf dcol(col, ncol * 4), doff(offset, (size_t)dg * 2 * kh * kw * Ho * Wo * 4), dg_im(nullptr, (size_t)C * H * W * 4, true);
  const int n = (int)ncol;
  mxnet::op::deformable_col2im_gpu_kernel<float><<<cuda_get_num_blocks(n), kThreads>>>(
      n, dcol.as<float>(), doff.as<float>(), C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, C / dg, Ho, Wo, dg_im.as<float>(), mxnet::kWriteTo);
  return finish(grad_im, dg_im, (size_t)C * H * W * 4);
}

// col, data_im, offset -> grad_offset [dg*2*kh*kw, Ho, Wo]
int ref_deformable_col2im_coord(const float* col, const float* im, const float* offset, int C, int H, int W, int kh, int kw, int ph,
                                int pw, int sh, int sw, int dh, int dw, int dg, float* grad_offset) {
  const int Ho = out_dim(H, kh, ph, sh, dh), Wo = out_dim(W, kw, pw, sw, dw);
  const size_t ncol = (size_t)C * kh * kw * Ho * Wo, noff = (size_t)dg * 2 * kh * kw * Ho * Wo;
  DevBuf dcol(col, ncol * 4), dim(im, (size_t)C * H * W * 4), doff(offset, noff * 4), dgo(nullptr, noff * 4, true);
  const int n = (int)noff;
  mxnet::op::deformable_col2im_coord_gpu_kernel<float><<<cuda_get_num_blocks(n), kThreads>>>(
      n, dcol.as<float>(), dim.as<float>(), doff.as<float>(), C, H, W, kh, kw, ph, pw, sh, sw, dh, dw, C * kh * kw / dg, Ho, Wo,
      dgo.as<float>(), mxnet::kWriteTo);
  return finish(grad_offset, dgo, noff * 4);
}

// data [N,C,H,W], rois [R,5], trans [R, 2*num_classes, part, part] (NULL with no_trans) -> top_data / top_count [R, output_dim, P, P]
int ref_psroi_forward(const float* data, const float* rois, const float* trans, int N, int C, int H, int W, int R, int no_trans,
                      float spatial_scale, int output_dim, int group_size, int pooled, int part, int sample_per_part, float trans_std,
                      int num_classes, float* top_data, float* top_count) {
  const size_t nout = (size_t)R * output_dim * pooled * pooled;
  DevBuf dd(data, (size_t)N * C * H * W * 4), dr(rois, (size_t)R * 5 * 4),
      dt(no_trans ? nullptr : trans, no_trans ? 4 : (size_t)R * 2 * num_classes * part * part * 4), dtop(nullptr, nout * 4, true), dcnt(nullptr, nout * 4, true);
  const int count = (int)nout, ncls = no_trans ? 1 : num_classes, cec = no_trans ? output_dim : output_dim / ncls;
  mshadow::cuda::DeformablePSROIPoolForwardKernel<float><<<cuda_get_num_blocks(count), kThreads>>>(
      count, dd.as<float>(), spatial_scale, C, H, W, pooled, pooled, dr.as<float>(), no_trans ? nullptr : dt.as<float>(), no_trans != 0,
      trans_std, sample_per_part, output_dim, group_size, part, ncls, cec, dtop.as<float>(), dcnt.as<float>());
  if (finish(top_data, dtop, nout * 4)) return -1;
  return hipMemcpy(top_count, dcnt.p, nout * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}

// -> in_grad [N,C,H,W], trans_grad [R, 2*num_classes, part, part] (both accumulated onto zeros with atomicAdd)
int ref_psroi_backward(const float* top_diff, const float* top_count, const float* data, const float* rois, const float* trans, int N,
                       int C, int H, int W, int R, int no_trans, float spatial_scale, int output_dim, int group_size, int pooled,
                       int part, int sample_per_part, float trans_std, int num_classes, float* in_grad, float* trans_grad) {
  const size_t nout = (size_t)R * output_dim * pooled * pooled, nin = (size_t)N * C * H * W;
  const size_t ntr = no_trans ? 1 : (size_t)R * 2 * num_classes * part * part;
  DevBuf ddiff(top_diff, nout * 4), dcnt(top_count, nout * 4), dd(data, nin * 4), dr(rois, (size_t)R * 5 * 4),
      dt(no_trans ? nullptr : trans, ntr * 4), dgi(nullptr, nin * 4, true), dgt(nullptr, ntr * 4, true);
  const int count = (int)nout, ncls = no_trans ? 1 : num_classes, cec = no_trans ? output_dim : output_dim / ncls;
  mshadow::cuda::DeformablePSROIPoolBackwardAccKernel<float><<<cuda_get_num_blocks(count), kThreads>>>(
      count, ddiff.as<float>(), dcnt.as<float>(), R, spatial_scale, C, H, W, pooled, pooled, output_dim, dgi.as<float>(),
      no_trans ? nullptr : dgt.as<float>(), dd.as<float>(), dr.as<float>(), no_trans ? nullptr : dt.as<float>(), no_trans != 0, trans_std,
      sample_per_part, group_size, part, ncls, cec);
  if (finish(in_grad, dgi, nin * 4)) return -1;
  if (!no_trans && hipMemcpy(trans_grad, dgt.p, ntr * 4, hipMemcpyDeviceToHost) != hipSuccess) return -2;
  return 0;
}

// the reference's own host function (lib/nms/nms_kernel.cu:80-144): boxes [n,5] (x1,y1,x2,y2,score) SORTED by score
int ref_nms(int* keep_out, int* num_out, const float* boxes, int n, float thresh) {
  _nms(keep_out, num_out, boxes, n, 5, thresh, 0);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // extern "C"
