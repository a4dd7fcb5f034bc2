// Build-time shim for oracle/_ref (TEST INFRASTRUCTURE ONLY -- never part of the product library).
//
// The reference's CUDA sources for this path are compiled UNEDITED, from where they lie under /root/reference, by
// oracle/build_ref.py with hipcc for gfx950:
//     relation_rcnn/operator_cxx/nn/deformable_im2col.cuh      (deformable_im2col / col2im / col2im_coord kernels)
//     relation_rcnn/operator_cxx/deformable_psroi_pooling.cu   (DeformablePSROIPool forward / backward kernels)
//     lib/nms/nms_kernel.cu                                    (nms_kernel + the host function _nms)
// Those files include MXNet / mshadow / dmlc headers that are not in /root/reference (MXNet 1.1.0 is an external
// dependency of the reference).  This header declares the handful of names the three files mention -- nothing of the
// algorithms: the kernels that run are the reference's own lines.  Constants follow mshadow 1.1 (mshadow/cuda/tensor_gpu-inl.cuh:
// kBaseThreadBits = 8, kMaxGridNum = 65535) and mxnet_op.h (cuda_get_num_blocks), CUDA_KERNEL_LOOP is MXNet's
// common/cuda_utils.h grid-stride loop.
#ifndef RELNET_ORACLE_REFSHIM_H_
#define RELNET_ORACLE_REFSHIM_H_
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <vector>

// ---- CUDA runtime names used by lib/nms/nms_kernel.cu and the psroi wrappers -> HIP runtime
#define cudaError_t hipError_t
#define cudaSuccess hipSuccess
#define cudaGetErrorString hipGetErrorString
#define cudaGetDevice hipGetDevice
#define cudaSetDevice hipSetDevice
#define cudaMalloc hipMalloc
#define cudaFree hipFree
#define cudaMemcpy hipMemcpy
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaPeekAtLastError hipPeekAtLastError
#define cudaStream_t hipStream_t

// ---- dmlc logging macros (only ever expanded inside wrappers that are not instantiated)
struct RefShimSink { template <typename T> RefShimSink& operator<<(const T&) { return *this; } };
#define CHECK_LT(a, b) RefShimSink()
#define CHECK_EQ(a, b) RefShimSink()
#define LOG(x) RefShimSink()
#define MSHADOW_CUDA_POST_KERNEL_CHECK(x)
#define MSHADOW_REAL_TYPE_SWITCH(dtype, DType, ...)

#ifndef CUDA_KERNEL_LOOP       // mxnet common/cuda_utils.h
#define CUDA_KERNEL_LOOP(i, n) \
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += blockDim.x * gridDim.x)
#endif

namespace mshadow {
typedef unsigned index_t;
struct gpu {};
struct cpu {};
template <typename Device> struct Stream { static hipStream_t GetStream(Stream<Device>*) { return 0; } };
template <int dim> struct Shape { index_t shape_[dim]; size_t Size() const { size_t s = 1; for (int i = 0; i < dim; ++i) s *= shape_[i]; return s; } };
template <typename Device, int dim, typename DType> struct Tensor {
  DType* dptr_; Shape<dim> shape_; Stream<Device>* stream_;
  index_t size(int i) const { return shape_.shape_[i]; }
};
namespace cuda {
const int kBaseThreadBits = 8;
const int kBaseThreadNum = 1 << kBaseThreadBits;
const int kMaxGridNum = 65535;
}  // namespace cuda
}  // namespace mshadow

namespace mxnet {
using mshadow::gpu;
using mshadow::cpu;
using mshadow::index_t;
enum OpReqType { kNullOp, kWriteTo, kWriteInplace, kAddTo };
struct TShape {
  std::vector<index_t> d;
  index_t ndim() const { return (index_t)d.size(); }
  index_t operator[](int i) const { return d[i]; }
  index_t ProdShape(int a, int b) const { index_t s = 1; for (int i = a; i < b; ++i) s *= d[i]; return s; }
};
namespace op {
struct Operator {};
struct DeformablePSROIPoolingParam {};
template <typename xpu, typename DType> struct DeformablePSROIPoolingOp : Operator { explicit DeformablePSROIPoolingOp(DeformablePSROIPoolingParam) {} };
template <typename xpu> Operator* CreateOp(DeformablePSROIPoolingParam param, int dtype);
namespace mxnet_op {
inline int cuda_get_num_blocks(const int N) {
  using namespace mshadow::cuda;
  return std::min(kMaxGridNum, (N + kBaseThreadNum - 1) / kBaseThreadNum);
}
}  // namespace mxnet_op
}  // namespace op
}  // namespace mxnet
#endif
