// Training losses of the relation-network graphs (SURVEY.md section 8, row A10 "losses"): forward value and the
// gradient with respect to the network output, i.e. the tensors the backward pass starts from.
//
//  * relnet_softmax_output   mx.sym.SoftmaxOutput(normalization='valid', use_ignore, ignore_label, multi_output,
//                            grad_scale): rpn_cls_prob (symbols/..._learn_nms.py:272-273) and cls_prob (:372-373,
//                            :379).  forward = softmax over the class axis; backward = (p - onehot(label)) *
//                            grad_scale / #valid, zero rows where label == ignore_label (MXNet v1.1.0
//                            src/operator/softmax_output-inl.h semantics).
//  * relnet_smooth_l1_loss   weight * mx.sym.smooth_l1(scalar=sigma, data=pred - target) wrapped in
//                            mx.sym.MakeLoss(grad_scale) (:276-278, :374-377): forward = the loss tensor,
//                            backward = grad_scale * weight * smooth_l1'(pred - target).
//  * relnet_nms_loss         nms_pos_loss / nms_neg_loss (:536-551): -t log(s + eps) k and -(1 - t) log(1 - s + eps) k
//                            with k = nms_loss_scale / (first_n * num_thresh); MakeLoss grad_scale = nms_pos_scale on
//                            the positive term.
#include "common.h"

namespace relnet {

struct SoftmaxOutArgs {
  const float* data;      // logical [outer, C, inner], element (o, c, i) at o*C*inner + c*inner + i
  const float* label;     // [outer, inner] class index as float (MXNet labels are float32)
  float* prob;            // same layout as data
  float* grad;            // same layout as data, or nullptr
  int* valid_count;       // device scratch [groups]: number of labels != ignore_label per normalisation group
  long outer, inner;
  long group;             // positions (outer * inner index) per normalisation group: 'valid' normalises per IMAGE in the
                          // reference (one image per executor); a batched call passes the positions of one image
  int C, use_ignore;
  float ignore_label, grad_scale;
};

__global__ __launch_bounds__(256) void softmax_count_kernel(SoftmaxOutArgs g) {
  const long total = g.outer * g.inner;
  // whole wavefronts walk the positions (an index past `total` only skips its own contribution)
  for (long base = (long)blockIdx.x * 256 + (threadIdx.x & ~63); base < total; base += (long)gridDim.x * 256) {
    const long idx = base + (threadIdx.x & 63);
    const bool in = idx < total;
    int local = (in && (!g.use_ignore || g.label[idx] != g.ignore_label)) ? 1 : 0;
    const long grp = (in ? idx : total - 1) / g.group;
    const long g0 = __shfl(grp, 0);
    if (__all(grp == g0)) {                       // the usual case: one normalisation group per wavefront -> one atomic
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) local += __shfl_xor(local, o);
      if ((threadIdx.x & 63) == 0 && local) atomicAdd(g.valid_count + g0, local);
    } else if (local) {
      atomicAdd(g.valid_count + grp, 1);
    }
  }
}

// one thread per (outer, inner) position; C <= a few hundred
__global__ __launch_bounds__(256) void softmax_output_kernel(SoftmaxOutArgs g) {
  const long total = g.outer * g.inner;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long o = idx / g.inner, i = idx % g.inner;
    const float gs = g.grad ? g.grad_scale / (float)max(g.valid_count[idx / g.group], 1) : 0.f;
    const float* x = g.data + o * g.C * g.inner + i;
    float m = x[0];
    for (int c = 1; c < g.C; ++c) m = fmaxf(m, x[(long)c * g.inner]);
    float s = 0.f;
    for (int c = 0; c < g.C; ++c) s += expf(x[(long)c * g.inner] - m);
    const float inv = 1.f / s;
    const float lab = g.label ? g.label[idx] : 0.f;
    const bool ignored = g.use_ignore && lab == g.ignore_label;
    const int li = (int)lab;
    float* p = g.prob + o * g.C * g.inner + i;
    float* d = g.grad ? g.grad + o * g.C * g.inner + i : nullptr;
    for (int c = 0; c < g.C; ++c) {
      const float pc = expf(x[(long)c * g.inner] - m) * inv;
      p[(long)c * g.inner] = pc;
      if (d) d[(long)c * g.inner] = ignored ? 0.f : (pc - (c == li ? 1.f : 0.f)) * gs;
    }
  }
}

struct SmoothL1Args {
  const float* pred; const float* target; const float* weight;
  float* loss; float* grad;      // either may be nullptr
  long n;
  float sigma, grad_scale;
};

__global__ __launch_bounds__(256) void smooth_l1_loss_kernel(SmoothL1Args g) {
  const float s2 = g.sigma * g.sigma, inv = 1.f / s2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < g.n; i += (long)gridDim.x * 256) {
    const float x = g.pred[i] - g.target[i], w = g.weight ? g.weight[i] : 1.f;
    const float ax = fabsf(x);
    // mshadow_op::smooth_l1_loss / smooth_l1_gradient: quadratic inside |x| < 1/sigma^2
    const bool quad = ax < inv;
    if (g.loss) g.loss[i] = w * (quad ? 0.5f * x * x * s2 : ax - 0.5f * inv);
    if (g.grad) g.grad[i] = g.grad_scale * w * (quad ? s2 * x : (x > 0.f ? 1.f : -1.f));
  }
}

struct NmsLossArgs {
  const float* score; const float* target;
  float* pos_loss; float* neg_loss; float* grad;
  long n;
  float eps, k, pos_scale;
};

__global__ __launch_bounds__(256) void nms_loss_kernel(NmsLossArgs g) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < g.n; i += (long)gridDim.x * 256) {
    const float s = g.score[i], t = g.target[i];
    const float a = s + g.eps, b = 1.f - s + g.eps;
    if (g.pos_loss) g.pos_loss[i] = g.k * (-(t * logf(a)));
    if (g.neg_loss) g.neg_loss[i] = g.k * (-((1.f - t) * logf(b)));
    if (g.grad) g.grad[i] = g.k * (g.pos_scale * (-t / a) + (1.f - t) / b);
  }
}

static inline unsigned grid1d(long n) {
  long b = (n + 255) / 256;
  return (unsigned)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_softmax_output_ex(const float* data, const float* label, float* prob, float* grad,
                                        int* valid_count_scratch, long outer, int C, long inner, int use_ignore,
                                        float ignore_label, float grad_scale, long group_positions, void* stream) {
  RELNET_REQUIRE(data && prob, "relnet_softmax_output: null operand");
  RELNET_REQUIRE(outer > 0 && C > 0 && inner > 0, "relnet_softmax_output: bad shape");
  RELNET_REQUIRE(!grad || (label && valid_count_scratch), "relnet_softmax_output: the gradient needs label and the count scratch");
  const long total = outer * inner;
  if (group_positions <= 0) group_positions = total;
  RELNET_REQUIRE(total % group_positions == 0, "relnet_softmax_output: %ld positions do not split into groups of %ld", total, group_positions);
  SoftmaxOutArgs g{data, label, prob, grad, valid_count_scratch, outer, inner, group_positions, C, use_ignore, ignore_label, grad_scale};
  hipStream_t s = (hipStream_t)stream;
  if (grad) {
    if (hipMemsetAsync(valid_count_scratch, 0, sizeof(int) * (size_t)(total / group_positions), s) != hipSuccess) { set_error("relnet_softmax_output: memset failed"); return -2; }
    softmax_count_kernel<<<grid1d(outer * inner), 256, 0, s>>>(g);
  }
  softmax_output_kernel<<<grid1d(outer * inner), 256, 0, s>>>(g);
  return check_launch("relnet_softmax_output");
}

extern "C" int relnet_softmax_output(const float* data, const float* label, float* prob, float* grad,
                                     int* valid_count_scratch, long outer, int C, long inner, int use_ignore,
                                     float ignore_label, float grad_scale, void* stream) {
  return relnet_softmax_output_ex(data, label, prob, grad, valid_count_scratch, outer, C, inner, use_ignore, ignore_label,
                                  grad_scale, 0, stream);
}

extern "C" int relnet_smooth_l1_loss(const float* pred, const float* target, const float* weight, float* loss,
                                     float* grad, long n, float sigma, float grad_scale, void* stream) {
  RELNET_REQUIRE(pred && target && (loss || grad), "relnet_smooth_l1_loss: null operand");
  RELNET_REQUIRE(n > 0 && sigma > 0.f, "relnet_smooth_l1_loss: bad n / sigma");
  SmoothL1Args g{pred, target, weight, loss, grad, n, sigma, grad_scale};
  smooth_l1_loss_kernel<<<grid1d(n), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_smooth_l1_loss");
}

extern "C" int relnet_nms_loss(const float* score, const float* target, float* pos_loss, float* neg_loss,
                               float* grad, long n, float eps, float loss_scale_over_normalizer, float pos_scale,
                               void* stream) {
  RELNET_REQUIRE(score && target && (pos_loss || neg_loss || grad), "relnet_nms_loss: null operand");
  RELNET_REQUIRE(n > 0, "relnet_nms_loss: bad n");
  NmsLossArgs g{score, target, pos_loss, neg_loss, grad, n, eps, loss_scale_over_normalizer, pos_scale};
  nms_loss_kernel<<<grid1d(n), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_nms_loss");
}
