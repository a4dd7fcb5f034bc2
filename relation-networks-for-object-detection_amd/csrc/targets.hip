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
  const int* num_rois;    // optional [B]: only the first num_rois[b] of the N input rows are proposals, the rest padding
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
  if (r >= g.N + G || (g.num_rois && r < g.N && r >= g.num_rois[b])) {   // padding row (fewer than Gmax gt boxes / than N proposals)
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
// ---------------------------------------------------------------------------------------
// RPN anchor targets on the device (reference: lib/rpn/rpn.py:80-244 assign_anchor, host numpy in the data loader).
//   anchors   base[a] + (x, y, x, y) * stride in (y, x, a) order (:127-141), float64 like numpy
//   inside    x1, y1 >= -border, x2 < im_w + border, y2 < im_h + border (:144-147): every other anchor keeps label -1
//   overlaps  bbox_overlaps (lib/bbox/bbox.pyx:33-55, float64); argmax over gt = first maximum (:166-167)
//   labels    max < RPN_NEGATIVE_OVERLAP -> 0; every anchor that ties some gt's best overlap -> 1 (:168-170,177);
//             max >= RPN_POSITIVE_OVERLAP -> 1 (:180); RPN_CLOBBER_POSITIVES moves the first rule last (:172-184)
//   sampling  more than num_fg positives / more than batch - #fg negatives: a uniformly random subset is switched to -1
//             (npr.choice, :189-204).  Here the random subset = the anchors with the SMALLEST keys, key = a 32-bit hash of
//             (seed, image, anchor index): reproducible, order independent, no RNG stream to keep in step
//   targets   bbox_transform(anchor, matched gt) in float64, stored float32 (:206-208); weights 1 on the positives (:210-211)
//   layouts   label [B, A*fh*fw] in (a, y, x) order, bbox_target / bbox_weight [B, 4A, fh, fw] (:236-239)
// Three launches: best overlap per gt (atomic max on the float64 bit pattern: overlaps are >= 0), labels + targets,
// sub-sampling (one workgroup per image, radix select over the 64-bit (key, index) words).
// ---------------------------------------------------------------------------------------
#pragma clang fp contract(off)
struct AnchorArgs {
  const float* gt;              // [B, Gmax, 5]
  const int* num_gt;            // [B]
  const float* im_info;         // [B, 3] (height, width, scale)
  double base[32][4];           // base anchors (A <= 32)
  float* label;                 // [B, A*fh*fw]
  float* bbox_target;           // [B, 4A, fh, fw]
  float* bbox_weight;           // [B, 4A, fh, fw]
  float* label_all;             // optional [B, A*fh*fw]: the labels BEFORE sub-sampling
  unsigned long long* gt_best;  // workspace [B, Gmax], zeroed by the entry point
  int A, fh, fw, Gmax, stride, batch_size, num_fg, allowed_border, clobber;
  double neg_ov, pos_ov;
  unsigned long long seed;
  const unsigned long long* seed_dev;   // optional device word added to `seed` (a step counter that a captured graph can advance)
};

__device__ __forceinline__ unsigned int anchor_key(unsigned long long seed, int b, int idx) {
  // 32-bit avalanche hash (two multiply-xorshift rounds) of the (seed, image, anchor) triple
  unsigned int x = (unsigned int)idx * 0x9E3779B1u ^ (unsigned int)(seed & 0xffffffffu) ^ ((unsigned int)b * 0x85EBCA77u)
                   ^ (unsigned int)(seed >> 32) * 0xC2B2AE3Du;
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

__device__ __forceinline__ bool anchor_box(const AnchorArgs& g, int b, int idx, double* box) {
  const int a = idx % g.A, k = idx / g.A, x = k % g.fw, y = k / g.fw;
  box[0] = g.base[a][0] + (double)(x * g.stride); box[1] = g.base[a][1] + (double)(y * g.stride);
  box[2] = g.base[a][2] + (double)(x * g.stride); box[3] = g.base[a][3] + (double)(y * g.stride);
  const double im_h = (double)g.im_info[b * 3], im_w = (double)g.im_info[b * 3 + 1], bd = (double)g.allowed_border;
  return box[0] >= -bd && box[1] >= -bd && box[2] < im_w + bd && box[3] < im_h + bd;
}

__global__ __launch_bounds__(256) void anchor_best_kernel(AnchorArgs g) {
  const int b = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
  const int total = g.A * g.fh * g.fw, G = g.num_gt[b];
  if (idx >= total || G <= 0) return;
  double box[4];
  if (!anchor_box(g, b, idx, box)) return;
  const float* gtb = g.gt + (long)b * g.Gmax * 5;
  for (int k = 0; k < G; ++k) {
    const double q[4] = {gtb[k * 5], gtb[k * 5 + 1], gtb[k * 5 + 2], gtb[k * 5 + 3]};
    const double ov = iou64(box, q);
    if (ov > 0.0) atomicMax(g.gt_best + (long)b * g.Gmax + k, (unsigned long long)__double_as_longlong(ov));
  }
}

__global__ __launch_bounds__(256) void anchor_label_kernel(AnchorArgs g) {
  const int b = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
  const int total = g.A * g.fh * g.fw, G = g.num_gt[b];
  if (idx >= total) return;
  const int a = idx % g.A, k = idx / g.A, x = k % g.fw, y = k / g.fw;
  double box[4];
  const bool inside = anchor_box(g, b, idx, box);
  float lab = -1.f;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  if (inside) {
    if (G <= 0) {
      lab = 0.f;                                                        // :186 no gt: everything inside is background
    } else {
      const float* gtb = g.gt + (long)b * g.Gmax * 5;
      double best = -1.0; int bi = 0; bool gt_arg = false;
      for (int kk = 0; kk < G; ++kk) {
        const double q[4] = {gtb[kk * 5], gtb[kk * 5 + 1], gtb[kk * 5 + 2], gtb[kk * 5 + 3]};
        const double ov = iou64(box, q);
        if (ov > best) { best = ov; bi = kk; }
        gt_arg = gt_arg || (ov == __longlong_as_double((long long)g.gt_best[(long)b * g.Gmax + kk]));
      }
      if (!g.clobber && best < g.neg_ov) lab = 0.f;
      if (gt_arg) lab = 1.f;
      if (best >= g.pos_ov) lab = 1.f;
      if (g.clobber && best < g.neg_ov) lab = 0.f;
      // bbox_transform.py:74-100 in float64 (float64 anchors, float32 gt promoted), rounded to float32 on store
      const float* q = gtb + bi * 5;
      const double ew = box[2] - box[0] + 1.0, eh = box[3] - box[1] + 1.0;
      const double ecx = box[0] + 0.5 * (ew - 1.0), ecy = box[1] + 0.5 * (eh - 1.0);
      const double gw = (double)q[2] - (double)q[0] + 1.0, gh = (double)q[3] - (double)q[1] + 1.0;
      const double gcx = (double)q[0] + 0.5 * (gw - 1.0), gcy = (double)q[1] + 0.5 * (gh - 1.0);
      t[0] = (float)((gcx - ecx) / (ew + 1e-14)); t[1] = (float)((gcy - ecy) / (eh + 1e-14));
      t[2] = (float)log(gw / ew); t[3] = (float)log(gh / eh);
    }
  }
  const long lo = (long)b * total + ((long)a * g.fh + y) * g.fw + x;     // (a, y, x) order
  g.label[lo] = lab;
  if (g.label_all) g.label_all[lo] = lab;
  const long hw = (long)g.fh * g.fw;
  float* bt = g.bbox_target + ((long)b * 4 * g.A + 4 * a) * hw + (long)y * g.fw + x;
#pragma unroll
  for (int c = 0; c < 4; ++c) bt[c * hw] = t[c];
}

// One workgroup per image.  which = 1: positives beyond num_fg, then which = 0: negatives beyond batch - #fg.
// The `drop` smallest 64-bit (key, index) words are found by an 8-bit radix select (8 rounds of a 256-bin LDS histogram).
__global__ __launch_bounds__(1024) void anchor_sample_kernel(AnchorArgs g) {
  __shared__ int s_hist[256];
  __shared__ int s_red[16];
  __shared__ int s_pick[2];                          // chosen digit, remaining rank
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int total = g.A * g.fh * g.fw;
  float* lab = g.label + (long)b * total;
  const unsigned long long seed = g.seed + (g.seed_dev ? *g.seed_dev : 0ull);
  const int hw = g.fh * g.fw;
  // position p (a, y, x order) <-> anchor index in (y, x, a) order, the index the keys are defined on
  auto key_of = [&](int p) -> unsigned long long {
    const int a = p / hw, k = p - a * hw, idx = k * g.A + a;
    return ((unsigned long long)anchor_key(seed, b, idx) << 32) | (unsigned int)idx;
  };
  auto block_count = [&](float want) -> int {
    int c = 0;
    for (int p = tid; p < total; p += 1024) c += lab[p] == want ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    __syncthreads();
    if (lane == 0) s_red[wave] = c;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < 16; ++w) t += s_red[w];
    return t;
  };
  for (int which = 1; which >= 0; --which) {
    const float want = (float)which;
    const int n = block_count(want);
    int keep = g.num_fg;
    if (which == 0) keep = g.batch_size - block_count(1.f);
    if (keep < 0) keep = 0;
    const int drop = n - keep;
    if (drop <= 0) continue;                          // (uniform: n and keep are block-wide values)
    unsigned long long prefix = 0ull, mask = 0ull;
    int need = drop;                                  // rank (1-based) of the threshold word among the words matching `prefix`
    for (int shift = 56; shift >= 0; shift -= 8) {
      if (tid < 256) s_hist[tid] = 0;
      __syncthreads();
      for (int p = tid; p < total; p += 1024)
        if (lab[p] == want) {
          const unsigned long long k = key_of(p);
          if ((k & mask) == prefix) atomicAdd(&s_hist[(int)((k >> shift) & 255ull)], 1);
        }
      __syncthreads();
      if (tid == 0) {                                 // first digit whose cumulative count reaches the rank
        int cum = 0, d = 0;
        for (; d < 256; ++d) {
          if (cum + s_hist[d] >= need) break;
          cum += s_hist[d];
        }
        s_pick[0] = d; s_pick[1] = need - cum;
      }
      __syncthreads();
      prefix |= (unsigned long long)s_pick[0] << shift;
      mask |= 255ull << shift;
      need = s_pick[1];
      __syncthreads();
    }
    // prefix = the drop-th smallest word (words are unique): everything <= it is switched off
    for (int p = tid; p < total; p += 1024)
      if (lab[p] == want && key_of(p) <= prefix) lab[p] = -1.f;
    __syncthreads();
  }
  // weights: RPN_BBOX_WEIGHTS (1, 1, 1, 1) on the surviving positives
  for (int p = tid; p < total; p += 1024) {
    const int a = p / hw, k = p - a * hw;
    const float w = lab[p] == 1.f ? 1.f : 0.f;
    float* bw = g.bbox_weight + ((long)b * 4 * g.A + 4 * a) * (long)hw + k;
#pragma unroll
    for (int c = 0; c < 4; ++c) bw[(long)c * hw] = w;
  }
}

#pragma clang fp contract(fast)
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

extern "C" int relnet_assign_anchor(const float* gt, const int* num_gt, const float* im_info, const double* base_anchors,
                                    float* label, float* bbox_target, float* bbox_weight, float* label_all,
                                    unsigned long long* workspace, int B, int A, int feat_h, int feat_w, int Gmax,
                                    int feat_stride, int rpn_batch_size, int num_fg, double negative_overlap,
                                    double positive_overlap, int clobber_positives, int allowed_border,
                                    unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
  RELNET_REQUIRE(gt && num_gt && im_info && base_anchors && label && bbox_target && bbox_weight && workspace,
                 "relnet_assign_anchor: null operand");
  RELNET_REQUIRE(B > 0 && A > 0 && A <= 32 && feat_h > 0 && feat_w > 0 && Gmax > 0, "relnet_assign_anchor: bad shape (A <= 32)");
  AnchorArgs g;
  g.gt = gt; g.num_gt = num_gt; g.im_info = im_info;
  for (int a = 0; a < A; ++a)
    for (int c = 0; c < 4; ++c) g.base[a][c] = base_anchors[a * 4 + c];
  g.label = label; g.bbox_target = bbox_target; g.bbox_weight = bbox_weight; g.label_all = label_all; g.gt_best = workspace;
  g.A = A; g.fh = feat_h; g.fw = feat_w; g.Gmax = Gmax; g.stride = feat_stride; g.batch_size = rpn_batch_size; g.num_fg = num_fg;
  g.allowed_border = allowed_border; g.clobber = clobber_positives; g.neg_ov = negative_overlap; g.pos_ov = positive_overlap;
  g.seed = seed; g.seed_dev = seed_dev;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(workspace, 0, (size_t)B * Gmax * sizeof(unsigned long long), s) != hipSuccess) {
    relnet::set_error("relnet_assign_anchor: memset failed");
    return -2;
  }
  const int total = A * feat_h * feat_w;
  dim3 grid((total + 255) / 256, B);
  anchor_best_kernel<<<grid, 256, 0, s>>>(g);
  anchor_label_kernel<<<grid, 256, 0, s>>>(g);
  anchor_sample_kernel<<<B, 1024, 0, s>>>(g);
  return check_launch("relnet_assign_anchor");
}

extern "C" int relnet_proposal_target_ex(const float* rois, const float* gt, const int* num_gt, float* rois_out,
                                         float* label, float* bbox_target, float* bbox_weight, int B, int N, int Gmax,
                                         int num_reg, int class_agnostic, float bg_thresh_hi, const double* means4,
                                         const double* stds4, const double* weights4, const int* num_rois, void* stream) {
  RELNET_REQUIRE(rois && gt && num_gt && rois_out && label && bbox_target && bbox_weight && means4 && stds4 && weights4,
                 "relnet_proposal_target: null operand");
  RELNET_REQUIRE(B > 0 && N >= 0 && Gmax >= 0 && N + Gmax > 0 && num_reg > 0, "relnet_proposal_target: bad shape");
  PTArgs g{rois, gt, num_gt, rois_out, label, bbox_target, bbox_weight, N, Gmax, num_reg, class_agnostic, bg_thresh_hi,
           {means4[0], means4[1], means4[2], means4[3]}, {stds4[0], stds4[1], stds4[2], stds4[3]},
           {weights4[0], weights4[1], weights4[2], weights4[3]}, num_rois};
  proposal_target_kernel<<<dim3((N + Gmax + 255) / 256, B), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_proposal_target");
}

extern "C" int relnet_proposal_target(const float* rois, const float* gt, const int* num_gt, float* rois_out,
                                      float* label, float* bbox_target, float* bbox_weight, int B, int N, int Gmax,
                                      int num_reg, int class_agnostic, float bg_thresh_hi, const double* means4,
                                      const double* stds4, const double* weights4, void* stream) {
  return relnet_proposal_target_ex(rois, gt, num_gt, rois_out, label, bbox_target, bbox_weight, B, N, Gmax, num_reg,
                                   class_agnostic, bg_thresh_hi, means4, stds4, weights4, nullptr, stream);
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
