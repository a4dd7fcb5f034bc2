// Detection post-processing on device (reference: relation_rcnn/core/tester.py:148-156
// im_detect decode, :244-268 per-class NMS / soft-NMS, :270-277 max_per_image; lib/nms/nms.py:
// 45-82 `nms`, :85-141 `soft_nms`).  The reference runs this part in numpy float64 on one host
// core (25 / 59 ms per image in its README); here every (image, class) pair is one wavefront.
//
//   detect_head_kernel      softmax over classes (SoftmaxActivation) + class-agnostic box
//                           decode (bbox_transform.py:103-140, float64) + clip + 1/scale.
//   class_nms_kernel        per (image, class): candidates with prob > thresh, then Gaussian
//                           soft-NMS (score *= exp(-iou^2/sigma), re-pick the max each step) or
//                           greedy NMS (drop iou > thresh); all arithmetic float64 like numpy.
//   image_topk_kernel       image-level score threshold = max_per_image-th largest score.
#include "common.h"

namespace relnet {

struct HeadArgs {
  const float* cls_score; long cs_ld;     // [R, C] logits
  const float* bbox_pred; long bp_ld;     // [R, 4*num_reg]; class-agnostic fg deltas at +4
  const float* rois;                      // [R, 5]
  const float* im_info;                   // [B, 3]
  float* cls_prob;                        // [R, C]
  double* boxes;                          // [R, 4] decoded, clipped, divided by im scale
  int R, C, rois_per_image, delta_off;
  const int* n_valid;                     // optional [B]: rows past n_valid[b] of image b are padding -> all-zero outputs
};

#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void detect_head_kernel(HeadArgs g) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (g.n_valid && (r % g.rois_per_image) >= g.n_valid[r / g.rois_per_image]) {     // padding row: never a detection
    for (int c = lane; c < g.C; c += 64) g.cls_prob[(long)r * g.C + c] = 0.f;
    if (lane < 4) g.boxes[(long)r * 4 + lane] = 0.0;
    return;
  }
  const float* z = g.cls_score + (long)r * g.cs_ld;
  float m = -INFINITY;
  for (int c = lane; c < g.C; c += 64) m = fmaxf(m, z[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.f;
  for (int c = lane; c < g.C; c += 64) s += expf(z[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  for (int c = lane; c < g.C; c += 64) g.cls_prob[(long)r * g.C + c] = expf(z[c] - m) / s;
  if (lane == 0) {
    const float* roi = g.rois + (long)r * 5;
    const float* d = g.bbox_pred + (long)r * g.bp_ld + g.delta_off;
    const float* info = g.im_info + (long)(r / g.rois_per_image) * 3;
    const double x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const double w = x2 - x1 + 1.0, h = y2 - y1 + 1.0;
    const double cx = x1 + 0.5 * (w - 1.0), cy = y1 + 0.5 * (h - 1.0);
    const double pcx = (double)d[0] * w + cx, pcy = (double)d[1] * h + cy;
    const double pw = (double)(float)exp((double)d[2]) * w, ph = (double)(float)exp((double)d[3]) * h;
    const double mx = (double)info[1] - 1.0, my = (double)info[0] - 1.0, sc = (double)info[2];
    double o[4] = {pcx - 0.5 * (pw - 1.0), pcy - 0.5 * (ph - 1.0), pcx + 0.5 * (pw - 1.0), pcy + 0.5 * (ph - 1.0)};
    o[0] = fmax(fmin(o[0], mx), 0.0); o[1] = fmax(fmin(o[1], my), 0.0);
    o[2] = fmax(fmin(o[2], mx), 0.0); o[3] = fmax(fmin(o[3], my), 0.0);
#pragma unroll
    for (int c = 0; c < 4; ++c) g.boxes[(long)r * 4 + c] = o[c] / sc;
  }
}

// ---------------------------------------------------------------------------------------
struct ClsNmsArgs {
  const float* cls_prob;     // [B, N, C]
  const double* boxes;       // [B, N, 4]
  double* dets;              // [B, C-1, N, 5]  x1,y1,x2,y2,score in pick order
  int* counts;               // [B, C-1]
  int N, C;
  float score_thresh;        // 1e-3 (tester.py:245)
  double nms_param;          // sigma (soft) or IoU threshold (hard)
  int soft;
  int max_picks;             // stop after this many picks per class (picks come out in
                             // non-increasing score order, so the first max_per_image picks of a
                             // class are the only ones that can survive tester.py:270-277)
  const double* scores64;    // optional [B, N] float64 scores of a SINGLE class (C == 2): used instead of cls_prob
                             // (the lib/nms/nms.py wrappers take float64 `dets`)
  int* pick_index;           // optional [B, C-1, N]: roi index of every pick (`keep` of nms.py:45-82)
  unsigned int* hist;        // PRUNE kernels: [B, kHistBins] zero-initialised by the caller: histogram of the keys of all picks of an image
  int top_k;                 // PRUNE kernels: max_per_image
};

// Image-level pruning of the per-class lists (tester.py:270-277 keeps, per image, the detections whose score is >= the top_k-th
// largest over all classes).  A class's pick sequence is non-increasing (soft-NMS only lowers scores, every pick is the current
// maximum), so once its LATEST pick lies below a lower bound of that final threshold, none of its later picks can be kept.
// The bound: every pick of the image is counted in a shared two-level histogram of its float64 score: 384 coarse bins (sign,
// exponent, 5 mantissa bits: 3 %) and, per coarse bin, 32 fine bins (the next 5 mantissa bits: 0.1 % -- with flat posteriors a class's
// pick sequence falls slowly, the coarse bin alone let 10-30 picks per class through).  (t, t2) = the coarse / fine bin in which the
// count from the top reaches top_k.  Histograms of SUBSETS of the final pick set can only give a lower (t, t2), so a stale or partial
// view (other classes still running, agent-scope relaxed reads) prunes less, never wrongly: a class stops when the bin pair of its
// latest pick is below (t, t2).  The lists written are prefixes of the full lists and contain every pick >= the final threshold, so
// relnet_image_topk returns the same detections; counts[] = picks actually produced.  With 80 similar classes (the benchmark's
// random-init heads: all 300 rois are candidates in every class) a class stops after a few picks instead of 100.
constexpr int kHistBins = 384;                      // scores in [2^-11, 2): (exponent - 1012) * 32 + 5 mantissa bits
constexpr int kHistFine = 32;                       // fine bins per coarse bin (mantissa bits 6..10)
constexpr int kHistWords = kHistBins * (1 + kHistFine);     // per image: [kHistBins] coarse, then [kHistBins][kHistFine]
__device__ __forceinline__ int score_bin(double sc, int& fine) {
  const long long bits = __double_as_longlong(sc);
  const long long bin = (bits >> 47) - (1012LL << 5);
  fine = (int)((bits >> 42) & (kHistFine - 1));
  if (bits <= 0 || bin < 0) { fine = 0; return 0; }
  if (bin >= kHistBins) { fine = kHistFine - 1; return kHistBins - 1; }
  return (int)bin;
}

// "absent / already picked / suppressed" marker of a candidate's score.  -inf, not -1: the lib/nms/nms.py twins accept any
// float64 dets[:, 4] (raw logits, negative scores), which must stay distinguishable from removed slots.
#define kGone (-INFINITY)

// Wave-wide maximum of a double without LDS traffic: four DPP steps inside each row of 16 lanes (quad_perm xor 1 / xor 2,
// row_half_mirror, row_mirror: the maximum is idempotent, mirrored partners are as good as a butterfly), then the four row values
// through v_readlane.  (__shfl_xor of a double is two ds_bpermute round trips per step; the per-pick arg-max chain of the
// class-NMS kernels is latency bound.)  The result is the same in every lane.
template <int CTRL> __device__ __forceinline__ double dpp_move_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_max_f64(double v) {
  v = fmax(v, dpp_move_f64<0xB1>(v));        // quad_perm [1,0,3,2]
  v = fmax(v, dpp_move_f64<0x4E>(v));        // quad_perm [2,3,0,1]
  v = fmax(v, dpp_move_f64<0x141>(v));       // row_half_mirror
  v = fmax(v, dpp_move_f64<0x140>(v));       // row_mirror
  return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// kPerLane * 64 >= N candidates per (image, class)
template <int kPerLane, bool PRUNE = false>
__global__ __launch_bounds__(64) void class_nms_kernel(ClsNmsArgs g) {
  const int cls = blockIdx.x + 1, b = blockIdx.y, lane = threadIdx.x;
  const float* prob = g.cls_prob + (long)b * g.N * g.C;
  const double* bx = g.boxes + (long)b * g.N * 4;
  double x1[kPerLane], y1[kPerLane], x2[kPerLane], y2[kPerLane], area[kPerLane], sc[kPerLane];
  // candidate slot s of lane l is roi index s*64 + l; sc == kGone marks "absent/picked"
  int n = 0;
#pragma unroll
  for (int s = 0; s < kPerLane; ++s) {
    const int i = s * 64 + lane;
    sc[s] = kGone;
    x1[s] = y1[s] = x2[s] = y2[s] = area[s] = 0.0;
    if (i < g.N) {
      const double p = g.scores64 ? g.scores64[(long)b * g.N + i] : (double)prob[(long)i * g.C + cls];
      if (p > (double)g.score_thresh) {
        sc[s] = p;
        x1[s] = bx[i * 4 + 0]; y1[s] = bx[i * 4 + 1]; x2[s] = bx[i * 4 + 2]; y2[s] = bx[i * 4 + 3];
        area[s] = (x2[s] - x1[s] + 1) * (y2[s] - y1[s] + 1);
        ++n;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  double* out = g.dets + (((long)b * (g.C - 1) + (cls - 1)) * g.N) * 5;
  int picked = 0;
  const int n_it = n < g.max_picks ? n : g.max_picks;
  for (int it = 0; it < n_it; ++it) {
    // arg-max over remaining; ties -> larger roi index (argsort()[::-1] convention): wave maximum, then the highest (slot, lane)
    // holding it -- ballots, all on the scalar unit
    double lbest = kGone;
#pragma unroll
    for (int s = 0; s < kPerLane; ++s) lbest = fmax(lbest, sc[s]);
    const double best = wave_max_f64(lbest);
    if (!(best > kGone)) break;                     // everything suppressed (hard NMS)
    int bi = -1;
#pragma unroll
    for (int s = kPerLane - 1; s >= 0; --s) {
      const unsigned long long m = __ballot(sc[s] == best);
      if (bi < 0 && m) bi = s * 64 + 63 - __clzll(m);
    }
    const int bl = bi & 63, bs = bi >> 6;
    double px1 = 0, py1 = 0, px2 = 0, py2 = 0, pa = 0;
#pragma unroll
    for (int s = 0; s < kPerLane; ++s)
      if (s == bs) { px1 = x1[s]; py1 = y1[s]; px2 = x2[s]; py2 = y2[s]; pa = area[s]; }
    px1 = readlane_f64(px1, bl); py1 = readlane_f64(py1, bl); px2 = readlane_f64(px2, bl); py2 = readlane_f64(py2, bl); pa = readlane_f64(pa, bl);
    if (lane == 0) {
      double* o = out + (long)picked * 5;
      o[0] = px1; o[1] = py1; o[2] = px2; o[3] = py2; o[4] = best;
      if (g.pick_index) g.pick_index[((long)b * (g.C - 1) + (cls - 1)) * g.N + picked] = bi;
    }
    ++picked;
    if constexpr (PRUNE) {
      // count this pick, then look at the image's histogram: bins [6 lane, 6 lane + 6), count from the top bin down
      unsigned int* hist = g.hist + (long)b * kHistWords;
      unsigned int* hist2 = hist + kHistBins;
      int myfine;
      const int mybin = score_bin(best, myfine);
      if (lane == 0) {
        __hip_atomic_fetch_add(hist + mybin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(hist2 + mybin * kHistFine + myfine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // (the look-up is an L2 round trip on the serial path of the class: every pick for the first four, then every fourth)
      if (picked <= 4 || !(picked & 3)) {
        unsigned int c[6], tot = 0;
#pragma unroll
        for (int q = 0; q < 6; ++q) { c[q] = __hip_atomic_load(hist + lane * 6 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); tot += c[q]; }
        unsigned int suf = tot;                        // inclusive suffix sum over lanes: picks in bins >= 6 lane
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned int v = __shfl_down(suf, o); if (lane + o < 64) suf += v; }
        const unsigned long long m = __ballot(suf >= (unsigned)g.top_k);
        if (m) {
          const int L = 63 - __clzll(m);               // highest lane whose suffix reaches top_k: the bound's coarse bin is one of its six
          unsigned int acc = suf - tot;                // picks in the lanes above
          int t = lane * 6;
          unsigned int above = acc;                    // picks in coarse bins above t
#pragma unroll
          for (int q = 5; q >= 0; --q) { if (acc + c[q] >= (unsigned)g.top_k) { t = lane * 6 + q; above = acc; break; } acc += c[q]; }
          t = __shfl(t, L); above = __shfl(above, L);
          if (mybin < t) break;                        // every later pick of this class is <= this one: below the image cut for good
          if (mybin == t) {
            // fine bins of coarse bin t: lane f < 32 holds fine bin f; count from the top fine bin down, on top of `above`
            unsigned int f = lane < kHistFine ? __hip_atomic_load(hist2 + t * kHistFine + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            unsigned int fs = f;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) { const unsigned int v = __shfl_down(fs, o); if (lane + o < 32) fs += v; }
            const unsigned long long m2 = __ballot(lane < kHistFine && above + fs >= (unsigned)g.top_k);
            // (the fine counts may lag the coarse count they refine: no lane qualifying means "no fine bound yet")
            if (m2 && myfine < 63 - __clzll(m2)) break;
          }
        }
      }
    }
#pragma unroll
    for (int s = 0; s < kPerLane; ++s) {
      if (s * 64 + lane == bi) { sc[s] = kGone; continue; }
      if (!(sc[s] > kGone)) continue;
      const double w = fmax(0.0, fmin(px2, x2[s]) - fmax(px1, x1[s]) + 1);
      const double h = fmax(0.0, fmin(py2, y2[s]) - fmax(py1, y1[s]) + 1);
      const double inter = w * h;
      if (inter > 0.0) {     // disjoint boxes: ovr = 0 -> weight exp(0) = 1 / never suppressed; skipping them is exact
        const double ovr = inter / (pa + area[s] - inter);
        if (g.soft) sc[s] = sc[s] * exp(-(ovr * ovr) / g.nms_param);     // nms.py:92
        else if (!(ovr <= g.nms_param)) sc[s] = kGone;                     // nms.py:79
      }
    }
  }
  if (lane == 0) g.counts[(long)b * (g.C - 1) + (cls - 1)] = picked;
}

// N up to 1024 candidates (FPN graphs, TOP_ROIS 1000): four wavefronts share one (image, class), 4 candidates per
// thread.  Same arithmetic and tie rule as the single-wave kernel above; the per-pick arg-max crosses the waves
// through LDS.  (16 candidates per lane in ONE wave needs 6 x 16 doubles per lane and spills thousands of VGPRs.)
template <int kPerThread>
__global__ __launch_bounds__(256) void class_nms_block_kernel(ClsNmsArgs g) {
  constexpr int NT = 256;
  __shared__ double s_best[4];
  __shared__ int s_bi[4];
  __shared__ double s_box[5];
  __shared__ int s_n;
  const int cls = blockIdx.x + 1, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* prob = g.cls_prob + (long)b * g.N * g.C;
  const double* bx = g.boxes + (long)b * g.N * 4;
  double x1[kPerThread], y1[kPerThread], x2[kPerThread], y2[kPerThread], area[kPerThread], sc[kPerThread];
  if (tid == 0) s_n = 0;
  __syncthreads();
  int n = 0;
#pragma unroll
  for (int s = 0; s < kPerThread; ++s) {
    const int i = s * NT + tid;
    sc[s] = kGone;
    x1[s] = y1[s] = x2[s] = y2[s] = area[s] = 0.0;
    if (i < g.N) {
      const double p = g.scores64 ? g.scores64[(long)b * g.N + i] : (double)prob[(long)i * g.C + cls];
      if (p > (double)g.score_thresh) {
        sc[s] = p;
        x1[s] = bx[i * 4 + 0]; y1[s] = bx[i * 4 + 1]; x2[s] = bx[i * 4 + 2]; y2[s] = bx[i * 4 + 3];
        area[s] = (x2[s] - x1[s] + 1) * (y2[s] - y1[s] + 1);
        ++n;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
  if (lane == 0) atomicAdd(&s_n, n);
  __syncthreads();
  n = s_n;
  double* out = g.dets + (((long)b * (g.C - 1) + (cls - 1)) * g.N) * 5;
  int picked = 0;
  const int n_it = n < g.max_picks ? n : g.max_picks;
  for (int it = 0; it < n_it; ++it) {
    double best = kGone; int bi = -1;
#pragma unroll
    for (int s = 0; s < kPerThread; ++s)
      if (sc[s] > best || (sc[s] == best && sc[s] > kGone)) { best = sc[s]; bi = s * NT + tid; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(bi, o);
      if (ob > best || (ob == best && oi > bi)) { best = ob; bi = oi; }
    }
    if (lane == 0) { s_best[wave] = best; s_bi[wave] = bi; }
    __syncthreads();
    best = s_best[0]; bi = s_bi[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
      const double ob = s_best[w]; const int oi = s_bi[w];
      if (ob > best || (ob == best && oi > bi)) { best = ob; bi = oi; }
    }
    if (!(best > kGone)) break;                     // uniform: every thread read the same LDS values
    if (tid == (bi % NT)) {
      const int bs = bi / NT;
#pragma unroll
      for (int s = 0; s < kPerThread; ++s)
        if (s == bs) { s_box[0] = x1[s]; s_box[1] = y1[s]; s_box[2] = x2[s]; s_box[3] = y2[s]; s_box[4] = area[s]; }
    }
    __syncthreads();
    const double px1 = s_box[0], py1 = s_box[1], px2 = s_box[2], py2 = s_box[3], pa = s_box[4];
    if (tid == 0) {
      double* o = out + (long)picked * 5;
      o[0] = px1; o[1] = py1; o[2] = px2; o[3] = py2; o[4] = best;
      if (g.pick_index) g.pick_index[((long)b * (g.C - 1) + (cls - 1)) * g.N + picked] = bi;
    }
    ++picked;
#pragma unroll
    for (int s = 0; s < kPerThread; ++s) {
      if (s * NT + tid == bi) { sc[s] = kGone; continue; }
      if (!(sc[s] > kGone)) continue;
      const double w = fmax(0.0, fmin(px2, x2[s]) - fmax(px1, x1[s]) + 1);
      const double h = fmax(0.0, fmin(py2, y2[s]) - fmax(py1, y1[s]) + 1);
      const double inter = w * h;
      if (inter > 0.0) {
        const double ovr = inter / (pa + area[s] - inter);
        if (g.soft) sc[s] = sc[s] * exp(-(ovr * ovr) / g.nms_param);
        else if (!(ovr <= g.nms_param)) sc[s] = kGone;
      }
    }
    __syncthreads();                           // s_best / s_box are rewritten in the next iteration
  }
  if (tid == 0) g.counts[(long)b * (g.C - 1) + (cls - 1)] = picked;
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------
struct ImgTopkArgs {
  const double* dets;   // [B, NC, N, 5]
  const int* counts;    // [B, NC]
  double* thresh;       // [B] image score threshold (-inf when total <= max_per_image)
  int* total;           // [B]
  float* out;           // [B, max_out, 6] class, score, x1,y1,x2,y2 (class-major, pick order)
  int* out_count;       // [B]
  int NC, N, max_per_image, max_out;
};

__device__ __forceinline__ unsigned long long dkey(double d) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(d);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}

constexpr int kTopkKeys = 6144;          // pick keys cached in LDS (48 KB); longer lists are re-read from global memory every pass

__global__ __launch_bounds__(1024) void image_topk_kernel(ImgTopkArgs g) {
  __shared__ unsigned int hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_total, s_out;
  __shared__ int s_off[129];                  // exclusive prefix of the class counts (the picks of an image as ONE list of `total` entries)
  __shared__ unsigned long long s_keys[kTopkKeys];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int* cnt = g.counts + (long)b * g.NC;
  const double* dets = g.dets + (long)b * g.NC * g.N * 5;
  if (tid == 0) {
    int t = 0;
    for (int c = 0; c < g.NC; ++c) { s_off[c] = t; t += cnt[c]; }
    s_off[g.NC] = t;
    s_total = t; s_out = 0;
  }
  __syncthreads();
  const int total = s_total;
  const bool cached = total <= kTopkKeys;
  // entry i of the list -> its key (the lists are short once relnet_class_nms_topk has pruned them: the selection below walks
  // `total` entries per pass, not NC x N slots)
  auto key_of = [&](int i) {
    int lo = 0, hi = g.NC;                    // last class whose offset is <= i
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
    return dkey(dets[((long)lo * g.N + (i - s_off[lo])) * 5 + 4]);
  };
  if (cached) {
    for (int i = tid; i < total; i += 1024) s_keys[i] = key_of(i);
    __syncthreads();
  }
  unsigned long long kth = 0ull;
  if (total > g.max_per_image) {
    if (tid == 0) { s_prefix = 0ull; s_remaining = g.max_per_image; }
    __syncthreads();
    for (int pass = 7; pass >= 0; --pass) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      const unsigned long long prefix = s_prefix;
      const int shift = pass * 8;
      for (int i = tid; i < total; i += 1024) {
        const unsigned long long key = cached ? s_keys[i] : key_of(i);
        if (pass == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
      }
      __syncthreads();
      if (tid == 0) {
        int rem = s_remaining, d = 255;
        for (; d > 0; --d) { if ((int)hist[d] >= rem) break; rem -= hist[d]; }
        s_remaining = rem;
        s_prefix = prefix | ((unsigned long long)d << shift);
      }
      __syncthreads();
    }
    kth = s_prefix;
  }
  // keep score >= threshold, class-major / pick order (tester.py:273-277)
  if (tid == 0) {
    g.total[b] = total;
    double th = -INFINITY;
    if (total > g.max_per_image) {
      const unsigned long long u = (kth >> 63) ? (kth & 0x7fffffffffffffffull) : ~kth;
      th = __longlong_as_double((long long)u);
    }
    g.thresh[b] = th;
  }
  // per-class kept counts -> offsets (keeps are a prefix of each pick-ordered class list)
  for (int c = tid; c < g.NC; c += 1024) {
    int kc = 0;
    for (int k = 0; k < cnt[c]; ++k) kc += (dkey(dets[((long)c * g.N + k) * 5 + 4]) >= kth) ? 1 : 0;
    s_off[c] = kc;
  }
  __syncthreads();
  if (tid == 0) {
    int acc = 0;
    for (int c = 0; c < g.NC; ++c) { const int kc = s_off[c]; s_off[c] = acc; acc += kc; }
    s_out = acc;
  }
  __syncthreads();
  for (int c = tid; c < g.NC; c += 1024) {
    int pos = s_off[c];
    for (int k = 0; k < cnt[c]; ++k) {
      const double* d = dets + ((long)c * g.N + k) * 5;
      if (dkey(d[4]) >= kth) {
        if (pos < g.max_out) {
          float* o = g.out + ((long)b * g.max_out + pos) * 6;
          o[0] = (float)(c + 1); o[1] = (float)d[4]; o[2] = (float)d[0]; o[3] = (float)d[1]; o[4] = (float)d[2]; o[5] = (float)d[3];
        }
        ++pos;
      }
    }
  }
  if (tid == 0) g.out_count[b] = s_out < g.max_out ? s_out : g.max_out;
}

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_detect_head_ex(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld,
                                     const float* rois, const float* im_info, float* cls_prob, double* boxes,
                                     int R, int C, int rois_per_image, int delta_off, const int* n_valid, void* stream) {
  RELNET_REQUIRE(cls_score && bbox_pred && rois && im_info && cls_prob && boxes, "relnet_detect_head: null operand");
  RELNET_REQUIRE(R > 0 && C > 1 && rois_per_image > 0, "relnet_detect_head: bad shape");
  HeadArgs g{cls_score, cs_ld, bbox_pred, bp_ld, rois, im_info, cls_prob, boxes, R, C, rois_per_image, delta_off, n_valid};
  detect_head_kernel<<<R, 64, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_detect_head");
}

extern "C" int relnet_detect_head(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld,
                                  const float* rois, const float* im_info, float* cls_prob, double* boxes,
                                  int R, int C, int rois_per_image, int delta_off, void* stream) {
  return relnet_detect_head_ex(cls_score, cs_ld, bbox_pred, bp_ld, rois, im_info, cls_prob, boxes, R, C, rois_per_image,
                               delta_off, nullptr, stream);
}

extern "C" int relnet_class_nms_ex(const float* cls_prob, const double* scores64, const double* boxes, double* dets,
                                   int* counts, int* pick_index, int B, int N, int C, float score_thresh,
                                   double nms_param, int soft, int max_picks, void* stream) {
  RELNET_REQUIRE((cls_prob || scores64) && boxes && dets && counts, "relnet_class_nms: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && N <= 1024 && C > 1, "relnet_class_nms: need 0 < N <= 1024 (N=%d)", N);
  RELNET_REQUIRE(!scores64 || C == 2, "relnet_class_nms: float64 scores are one foreground class (C == 2), got C=%d", C);
  ClsNmsArgs g{cls_prob, boxes, dets, counts, N, C, score_thresh, nms_param, soft, max_picks > 0 ? max_picks : N,
               scores64, pick_index, nullptr, 0};
  dim3 grid(C - 1, B);
  hipStream_t s = (hipStream_t)stream;
  if (N <= 320) class_nms_kernel<5><<<grid, 64, 0, s>>>(g);
  else if (N <= 512) class_nms_kernel<8><<<grid, 64, 0, s>>>(g);
  else class_nms_block_kernel<4><<<grid, 256, 0, s>>>(g);
  return check_launch("relnet_class_nms");
}

extern "C" int relnet_class_nms(const float* cls_prob, const double* boxes, double* dets, int* counts,
                                int B, int N, int C, float score_thresh, double nms_param, int soft,
                                int max_picks, void* stream) {
  return relnet_class_nms_ex(cls_prob, nullptr, boxes, dets, counts, nullptr, B, N, C, score_thresh, nms_param, soft,
                             max_picks, stream);
}

// relnet_class_nms with image-level pruning (see kHistBins above): dets / counts as relnet_class_nms, but a class list stops as
// soon as its next pick cannot be among the top_k scores of its image.  `hist`: B x relnet_class_nms_hist_bins() unsigned ints (coarse + fine),
// ZEROED by the caller before every call.  N <= 512.
extern "C" int relnet_class_nms_hist_bins(void) { return kHistWords; }
extern "C" int relnet_class_nms_topk(const float* cls_prob, const double* boxes, double* dets, int* counts, void* hist, int B, int N,
                                     int C, float score_thresh, double nms_param, int soft, int max_picks, int top_k, void* stream) {
  RELNET_REQUIRE(cls_prob && boxes && dets && counts && hist, "relnet_class_nms_topk: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && N <= 512 && C > 1 && top_k > 0, "relnet_class_nms_topk: need 0 < N <= 512, top_k > 0 (N=%d top_k=%d)", N, top_k);
  ClsNmsArgs g{cls_prob, boxes, dets, counts, N, C, score_thresh, nms_param, soft, max_picks > 0 ? max_picks : N, nullptr, nullptr,
               (unsigned int*)hist, top_k};
  dim3 grid(C - 1, B);
  if (N <= 320) class_nms_kernel<5, true><<<grid, 64, 0, (hipStream_t)stream>>>(g);
  else class_nms_kernel<8, true><<<grid, 64, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_class_nms_topk");
}

extern "C" int relnet_image_topk(const double* dets, const int* counts, double* thresh, int* total,
                                 float* out, int* out_count, int B, int NC, int N, int max_per_image,
                                 int max_out, void* stream) {
  RELNET_REQUIRE(dets && counts && thresh && total && out && out_count, "relnet_image_topk: null operand");
  RELNET_REQUIRE(B > 0 && NC > 0 && NC <= 128 && N > 0 && max_per_image > 0 && max_out >= max_per_image, "relnet_image_topk: bad shape");
  ImgTopkArgs g{dets, counts, thresh, total, out, out_count, NC, N, max_per_image, max_out};
  image_topk_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_image_topk");
}
