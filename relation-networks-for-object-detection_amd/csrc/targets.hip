// Training-target operators on device (reference: operator_py/proposal_target.py:44-93 ->
// core/rcnn.py:288-325 sample_rois_v2 -> lib/bbox/bbox.pyx:33-55, bbox_transform.py:74-100,
// bbox_regression.py:120-140; operator_py/box_annotator_ohem.py:26-53; operator_py/nms_multi_target.py:
// 24-74).  In the reference each of these is a numpy CustomOp behind an .asnumpy() host round trip
// (three of the four syncs per training step); here they are small integer / float64 kernels.
#include "common.h"

namespace relnet {

#pragma clang fp contract(off)
__device__ __forceinline__ double iou64(const double* a, const double* b) {
  // bbox.pyx:33-55: +1 extents, 0 when disjoint
  const double iw = fmin(a[2], b[2]) - fmax(a[0], b[0]) + 1.0;
  if (!(iw > 0)) return 0.0;
  const double ih = fmin(a[3], b[3]) - fmax(a[1], b[1]) + 1.0;
  if (!(ih > 0)) return 0.0;
  const double ua = (a[2] - a[0] + 1.0) * (a[3] - a[1] + 1.0) + (b[2] - b[0] + 1.0) * (b[3] - b[1] + 1.0) - iw * ih;
  return iw * ih / ua;
}

struct PTArgs {
  const float* rois;      // [B, N, 5]
  const float* gt;        // [B, Gmax, 5]  x1,y1,x2,y2,cls
  const int* num_gt;      // [B]
  float* rois_out;        // [B, N+Gmax, 5]
  float* label;           // [B, N+Gmax]      (-1 on rows past N + num_gt[b])
  float* bbox_target;     // [B, N+Gmax, 4*num_reg]
  float* bbox_weight;     // [B, N+Gmax, 4*num_reg]
  int N, Gmax, num_reg, class_agnostic;
  float bg_thresh_hi;
  double mean[4], stdv[4], bw[4];
};

__global__ __launch_bounds__(256) void proposal_target_kernel(PTArgs g) {
  const int b = blockIdx.y, r = blockIdx.x * 256 + threadIdx.x;
  const int R = g.N + g.Gmax;
  if (r >= R) return;
  const int G = g.num_gt[b];
  const float* gtb = g.gt + (long)b * g.Gmax * 5;
  float* ro = g.rois_out + ((long)b * R + r) * 5;
  float* bt = g.bbox_target + ((long)b * R + r) * 4 * g.num_reg;
  float* bwp = g.bbox_weight + ((long)b * R + r) * 4 * g.num_reg;
  for (int c = 0; c < 4 * g.num_reg; ++c) { bt[c] = 0.f; bwp[c] = 0.f; }
  if (r >= g.N + G) {                         // padding row (image has fewer than Gmax gt boxes)
    for (int c = 0; c < 5; ++c) ro[c] = 0.f;
    ro[0] = (float)b;                         // still a valid image index for the batched pooling kernels
    g.label[(long)b * R + r] = -1.f;
    return;
  }
  float box[4];
  if (r < g.N) {
    const float* p = g.rois + ((long)b * g.N + r) * 5;
    ro[0] = p[0];
    for (int c = 0; c < 4; ++c) box[c] = p[1 + c];
  } else {                                    // proposal_target.py:64-67: gt boxes appended with batch idx 0 -- the
    ro[0] = (float)b;                         // reference runs one image per executor; batched: the image's own index
    for (int c = 0; c < 4; ++c) box[c] = gtb[(r - g.N) * 5 + c];
  }
  for (int c = 0; c < 4; ++c) ro[1 + c] = box[c];
  float lab = 0.f;
  if (G > 0) {
    const double a[4] = {box[0], box[1], box[2], box[3]};
    double best = -1.0; int bi = 0;
    for (int k = 0; k < G; ++k) {
      const double q[4] = {gtb[k * 5], gtb[k * 5 + 1], gtb[k * 5 + 2], gtb[k * 5 + 3]};
      const double ov = iou64(a, q);
      if (ov > best) { best = ov; bi = k; }   // argmax: first maximum
    }
    lab = gtb[bi * 5 + 4];
    if (best < (double)g.bg_thresh_hi) lab = 0.f;                      // rcnn.py:309-310
    if (lab > 0.f) {
      // bbox_transform.py:74-100 on float32 arrays (numpy keeps float32), log correctly rounded
      const float* q = gtb + bi * 5;
      const float ew = box[2] - box[0] + 1.0f, eh = box[3] - box[1] + 1.0f;
      const float ecx = box[0] + 0.5f * (ew - 1.0f), ecy = box[1] + 0.5f * (eh - 1.0f);
      const float gw = q[2] - q[0] + 1.0f, gh = q[3] - q[1] + 1.0f;
      const float gcx = q[0] + 0.5f * (gw - 1.0f), gcy = q[1] + 0.5f * (gh - 1.0f);
      const float t[4] = {(gcx - ecx) / (ew + 1e-14f), (gcy - ecy) / (eh + 1e-14f),
                          (float)log((double)(gw / ew)), (float)log((double)(gh / eh))};
      const int start = g.class_agnostic ? 4 : 4 * (int)lab;          // bbox_regression.py:134-138
      for (int c = 0; c < 4; ++c) {
        bt[start + c] = (float)(((double)t[c] - g.mean[c]) / g.stdv[c]);
        bwp[start + c] = (float)g.bw[c];
      }
    }
  }
  g.label[(long)b * R + r] = lab;
}

// ---------------------------------------------------------------------------------------
struct OhemArgs {
  const float* cls_score;   // [B, R, C]
  const float* bbox_pred;   // [B, R, D]
  const float* labels;      // [B, R]
  const float* bbox_targets;// [B, R, D]
  const float* bbox_weights;// [B, R, D]
  float* labels_ohem;       // [B, R]
  float* weights_ohem;      // [B, R, D]
  float* loss;              // [B, R] optional (per-roi loss)
  int R, C, D, roi_per_img;
};

__device__ __forceinline__ unsigned int fkey_t(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void ohem_kernel(OhemArgs g) {
  __shared__ unsigned long long keys[2048];
  __shared__ unsigned short rank_of[2048];
  const int b = blockIdx.x, tid = threadIdx.x;
  int np2 = 1;
  while (np2 < g.R) np2 <<= 1;
  for (int r = tid; r < np2; r += 1024) {
    unsigned long long key = 0ull;
    if (r < g.R) {
      const float* z = g.cls_score + ((long)b * g.R + r) * g.C;
      const float lab = g.labels[(long)b * g.R + r];
      float tot = -INFINITY;                                     // ignored rows (label < 0) rank last
      if (lab >= 0.f) {
        float m = -INFINITY;
        for (int c = 0; c < g.C; ++c) m = fmaxf(m, z[c]);
        double s = 0.0;
        for (int c = 0; c < g.C; ++c) s += exp((double)(z[c] - m));
        const float p = (float)(exp((double)(z[(int)lab] - m)) / s) + 1e-14f;      // box_annotator_ohem.py:33-36
        const float lc = -(float)log((double)p);
        const float* bp = g.bbox_pred + ((long)b * g.R + r) * g.D;
        const float* t = g.bbox_targets + ((long)b * g.R + r) * g.D;
        const float* w = g.bbox_weights + ((long)b * g.R + r) * g.D;
        double lb = 0.0;
        for (int c = 0; c < g.D; ++c) {
          const float d = bp[c] - t[c];
          const float sl = fabsf(d) < 1.0f ? 0.5f * d * d : fabsf(d) - 0.5f;        // smooth_l1, sigma 1
          lb += (double)(w[c] * sl);
        }
        tot = lc + (float)lb;
      }
      if (g.loss) g.loss[(long)b * g.R + r] = tot;
      key = ((unsigned long long)fkey_t(tot) << 16) | (unsigned)r;               // ties: larger index first
    }
    keys[r] = key;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], d = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < d) : (a > d)) { keys[i] = d; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int i = tid; i < g.R; i += 1024) rank_of[(int)(keys[i] & 0xffffu)] = (unsigned short)i;
  __syncthreads();
  for (int r = tid; r < g.R; r += 1024) {
    const bool keep = rank_of[r] < g.roi_per_img;
    g.labels_ohem[(long)b * g.R + r] = keep ? g.labels[(long)b * g.R + r] : -1.f;
    for (int c = 0; c < g.D; ++c)
      g.weights_ohem[((long)b * g.R + r) * g.D + c] = keep ? g.bbox_weights[((long)b * g.R + r) * g.D + c] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------
struct NMTArgs {
  const float* bbox;      // [B, F, C, 4]
  const float* gt;        // [B, Gmax, 5]
  const int* num_gt;      // [B]
  const float* score;     // [B, F, C]
  float* out;             // [B, F, C, T]
  int F, C, T, Gmax;
  double thresh[8];
};

constexpr int kNmtF = 256;      // max first_n
constexpr int kNmtG = 128;      // max gt boxes per image

__global__ __launch_bounds__(kNmtF) void nms_multi_target_kernel(NMTArgs g) {
  __shared__ double sgt[kNmtG * 4];
  __shared__ int s_ng;
  __shared__ double red_v[kNmtF / 64];
  __shared__ int red_i[kNmtF / 64];
  __shared__ int s_best;
  const int c = blockIdx.x, b = blockIdx.y, r = threadIdx.x;
  const int lane = r & 63, wave = r >> 6;
  const float* gtb = g.gt + (long)b * g.Gmax * 5;
  if (r == 0) {
    int n = 0;
    for (int k = 0; k < g.num_gt[b]; ++k)
      if ((int)gtb[k * 5 + 4] == c + 1) {
        for (int q = 0; q < 4; ++q) sgt[n * 4 + q] = (double)gtb[k * 5 + q];
        ++n;
      }
    s_ng = n;
  }
  __syncthreads();
  const int ng = s_ng;
  const bool on = r < g.F;
  float* o = on ? g.out + (((long)b * g.F + r) * g.C + c) * g.T : nullptr;
  if (on) for (int t = 0; t < g.T; ++t) o[t] = 0.f;
  if (ng == 0) return;
  double box[4] = {0, 0, 0, 0};
  double sc = 0.0;
  if (on) {
    const float* p = g.bbox + (((long)b * g.F + r) * g.C + c) * 4;
    for (int q = 0; q < 4; ++q) box[q] = (double)p[q];
    sc = (double)g.score[((long)b * g.F + r) * g.C + c];
  }
  double best = -1.0; int amax = 0;
  for (int k = 0; k < ng; ++k) {
    const double ov = iou64(box, sgt + 4 * k);
    if (ov > best) { best = ov; amax = k; }
  }
  for (int t = 0; t < g.T; ++t) {
    const double th = g.thresh[t];
    const bool valid_row = on && (best > th);          // some gt overlaps > th (best is the row maximum)
    for (int k = 0; k < ng; ++k) {
      // overlap_score[r, k] = score * (ov > th) * (argmax == k); argmax over r = first maximum
      double v = 0.0;
      if (on && amax == k && iou64(box, sgt + 4 * k) > th) v = sc;
      double bv = on ? v : -1.0; int bi = r;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        const double ov2 = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov2 > bv || (ov2 == bv && oi < bi)) { bv = ov2; bi = oi; }
      }
      if (lane == 0) { red_v[wave] = bv; red_i[wave] = bi; }
      __syncthreads();
      if (r == 0) {
        double fv = red_v[0]; int fi = red_i[0];
        for (int w2 = 1; w2 < (g.F + 63) / 64; ++w2)
          if (red_v[w2] > fv || (red_v[w2] == fv && red_i[w2] < fi)) { fv = red_v[w2]; fi = red_i[w2]; }
        s_best = fi;
      }
      __syncthreads();
      if (on && r == s_best && valid_row) o[t] = 1.f;   // np.intersect1d(max_score_indices, valid_bbox_indices)
      __syncthreads();
    }
  }
}
#pragma clang fp contract(fast)

}  // namespace relnet

using namespace relnet;

// ---------------------------------------------------------------------------------------
// lib/bbox/bbox.pyx:15-55 `bbox_overlaps_cython(boxes f64 [N,4], query_boxes f64 [K,4]) -> f64 [N,K]`
// (callers: core/rcnn.py:303, operator_py/nms_multi_target.py:51, lib/rpn/rpn.py:163): one thread per (n, k).
namespace relnet {
struct OverlapArgs { const double* boxes; const double* query; double* out; int N, K; };
__global__ __launch_bounds__(256) void bbox_overlaps_kernel(OverlapArgs g) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)g.N * g.K) return;
  const int n = (int)(t / g.K), k = (int)(t - (long)n * g.K);
  g.out[t] = iou64(g.boxes + 4L * n, g.query + 4L * k);
}
}  // namespace relnet

extern "C" int relnet_bbox_overlaps(const double* boxes, const double* query_boxes, double* overlaps, int N, int K,
                                    void* stream) {
  RELNET_REQUIRE(N >= 0 && K >= 0, "relnet_bbox_overlaps: bad shape N=%d K=%d", N, K);
  if (N == 0 || K == 0) return 0;
  RELNET_REQUIRE(boxes && query_boxes && overlaps, "relnet_bbox_overlaps: null operand");
  relnet::OverlapArgs g{boxes, query_boxes, overlaps, N, K};
  const long total = (long)N * K;
  relnet::bbox_overlaps_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_bbox_overlaps");
}

extern "C" int relnet_proposal_target(const float* rois, const float* gt, const int* num_gt, float* rois_out,
                                      float* label, float* bbox_target, float* bbox_weight, int B, int N, int Gmax,
                                      int num_reg, int class_agnostic, float bg_thresh_hi, const double* means4,
                                      const double* stds4, const double* weights4, void* stream) {
  RELNET_REQUIRE(rois && gt && num_gt && rois_out && label && bbox_target && bbox_weight && means4 && stds4 && weights4,
                 "relnet_proposal_target: null operand");
  RELNET_REQUIRE(B > 0 && N >= 0 && Gmax >= 0 && N + Gmax > 0 && num_reg > 0, "relnet_proposal_target: bad shape");
  PTArgs g{rois, gt, num_gt, rois_out, label, bbox_target, bbox_weight, N, Gmax, num_reg, class_agnostic, bg_thresh_hi,
           {means4[0], means4[1], means4[2], means4[3]}, {stds4[0], stds4[1], stds4[2], stds4[3]},
           {weights4[0], weights4[1], weights4[2], weights4[3]}};
  proposal_target_kernel<<<dim3((N + Gmax + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_proposal_target");
}

extern "C" int relnet_box_annotator_ohem(const float* cls_score, const float* bbox_pred, const float* labels,
                                         const float* bbox_targets, const float* bbox_weights, float* labels_ohem,
                                         float* weights_ohem, float* loss, int B, int R, int C, int D,
                                         int roi_per_img, void* stream) {
  RELNET_REQUIRE(cls_score && bbox_pred && labels && bbox_targets && bbox_weights && labels_ohem && weights_ohem,
                 "relnet_box_annotator_ohem: null operand");
  RELNET_REQUIRE(B > 0 && R > 0 && R <= 2048 && C > 1 && D > 0, "relnet_box_annotator_ohem: need 0 < R <= 2048");
  OhemArgs g{cls_score, bbox_pred, labels, bbox_targets, bbox_weights, labels_ohem, weights_ohem, loss, R, C, D, roi_per_img};
  ohem_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_box_annotator_ohem");
}

extern "C" int relnet_nms_multi_target(const float* bbox, const float* gt, const int* num_gt, const float* score,
                                       float* out, int B, int F, int C, int Gmax, const double* thresh, int T,
                                       void* stream) {
  RELNET_REQUIRE(bbox && gt && num_gt && score && out && thresh, "relnet_nms_multi_target: null operand");
  RELNET_REQUIRE(B > 0 && F > 0 && F <= kNmtF && C > 0 && Gmax <= kNmtG && T > 0 && T <= 8,
                 "relnet_nms_multi_target: need first_n <= %d, gt <= %d, thresholds <= 8", kNmtF, kNmtG);
  NMTArgs g{};
  g.bbox = bbox; g.gt = gt; g.num_gt = num_gt; g.score = score; g.out = out; g.F = F; g.C = C; g.T = T; g.Gmax = Gmax;
  for (int t = 0; t < T; ++t) g.thresh[t] = thresh[t];
  nms_multi_target_kernel<<<dim3(C, B), kNmtF, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_nms_multi_target");
}
