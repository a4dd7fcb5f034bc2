// Detection post-processing on device (reference: relation_rcnn/core/tester.py:148-156
// im_detect decode, :244-268 per-class NMS / soft-NMS, :270-277 max_per_image; lib/nms/nms.py:
// 45-82 `nms`, :85-141 `soft_nms`).  The reference runs this part in numpy float64 on one host
// core (25 / 59 ms per image in its README); here every (image, class) pair is one workgroup, one roi per thread.
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
// pick sequence falls slowly, the coarse bin alone let 10-30 picks per class through).  A class stops when at least top_k picks of
// the image lie in STRICTLY higher (coarse, fine) bins than its latest pick: those picks have strictly higher scores, so the image
// threshold (the top_k-th largest score) is above everything this class can still produce.  Histograms of SUBSETS of the final pick
// set only undercount (other classes still running, agent-scope relaxed reads, fine counts lagging the coarse ones), so a stale or
// partial view prunes less, never wrongly.  The lists written are prefixes of the full lists and contain every pick >= the final
// threshold, so relnet_image_topk returns the same detections; counts[] = picks actually produced.  With 80 similar classes (the
// benchmark's random-init heads: all 300 rois are candidates in every class) a class stops after a few picks instead of 100.
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

// Wave-wide sum of an unsigned (same DPP pairing as wave_max_f64: every step adds two disjoint groups).
__device__ __forceinline__ unsigned int wave_sum_u32(unsigned int v) {
  v += (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xf, 0xf, false);
  v += (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xf, 0xf, false);
  v += (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xf, 0xf, false);
  v += (unsigned int)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xf, 0xf, false);
  return (unsigned int)__builtin_amdgcn_readlane((int)v, 0) + (unsigned int)__builtin_amdgcn_readlane((int)v, 16) +
         (unsigned int)__builtin_amdgcn_readlane((int)v, 32) + (unsigned int)__builtin_amdgcn_readlane((int)v, 48);
}

// One workgroup of WAVES wavefronts per (image, class), PER candidates per thread (candidate s of thread t = roi s * 64 WAVES + t).
// The per-pick chain is the serial path of the whole post-processing (a class that owns k of the image's max_per_image
// detections makes k + 1 picks one after the other), so it is kept short:
//   before the barrier  each wave: maximum of its 64 scores (DPP), highest lane holding it (ballot: ties -> larger roi index, the
//                       argsort()[::-1] convention of nms.py), that lane's box through v_readlane -> one LDS slot per wave;
//   after the barrier   every thread reads the WAVES slots (LDS broadcasts), takes the best (ties -> later wave = larger roi
//                       index) and rescores its own candidate: one float64 division + exp per thread instead of five per lane
//                       in the one-wave form of rounds 1-3 (2.7 us per pick there, the fp64 exp chains of five slots back to back).
// One barrier per pick: the slots are double buffered by pick parity.
// PRUNE: wave 0 counts the pick in the image's histogram and at once requests the counts it needs for the NEXT decision
// (6 coarse bins per lane + the fine bins of the pick's coarse bin); they land while the candidates are rescored and are consumed
// before the next barrier.  The class stops when at least top_k picks of the image lie in strictly higher (coarse, fine) bins than
// its latest pick: A = picks in coarse bins above, F = picks in higher fine bins of the same coarse bin, stop <=> A + F >= top_k
// (one masked wave sum).  The flag travels through LDS, so the whole workgroup leaves the loop at the same pick.
template <int WAVES, int PER, bool PRUNE>
__global__ __launch_bounds__(64 * WAVES) void class_nms_kernel(ClsNmsArgs g) {
  constexpr int NT = 64 * WAVES;
  struct Slot { double best, x1, y1, x2, y2, area; int bi, pad; };
  __shared__ Slot s_slot[2][WAVES];
  __shared__ int s_n[WAVES];
  __shared__ int s_stop[2];
  const int cls = blockIdx.x + 1, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double x1[PER], y1[PER], x2[PER], y2[PER], area[PER], sc[PER];      // candidate s of thread t is roi s NT + t; sc == kGone: absent / picked / suppressed
  int n = 0;
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const int i = s * NT + tid;
    sc[s] = kGone;
    x1[s] = y1[s] = x2[s] = y2[s] = area[s] = 0.0;
    if (i < g.N) {
      const double p = g.scores64 ? g.scores64[(long)b * g.N + i] : (double)g.cls_prob[((long)b * g.N + i) * g.C + cls];
      if (p > (double)g.score_thresh) {
        const double* bx = g.boxes + ((long)b * g.N + i) * 4;
        sc[s] = p;
        x1[s] = bx[0]; y1[s] = bx[1]; x2[s] = bx[2]; y2[s] = bx[3];
        area[s] = (x2[s] - x1[s] + 1) * (y2[s] - y1[s] + 1);
      }
    }
    n += __popcll(__ballot(sc[s] > kGone));
  }
  if constexpr (WAVES > 1) {
    if (lane == 0) s_n[wave] = n;
    if (tid < 2) s_stop[tid] = 0;
    __syncthreads();
    n = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) n += s_n[w];
  }
  double* out = g.dets + (((long)b * (g.C - 1) + (cls - 1)) * g.N) * 5;
  unsigned int* hist = PRUNE ? g.hist + (long)b * kHistWords : nullptr;
  int picked = 0;
  const int n_it = n < g.max_picks ? n : g.max_picks;
  unsigned int hc[6] = {0u, 0u, 0u, 0u, 0u, 0u}, hf = 0u;       // wave 0: the counts requested after the previous pick
  int mybin = 0, myfine = 0;
  for (int it = 0; it < n_it; ++it) {
    const int par = it & 1;
    // this wave's best candidate: maximum (DPP), then the highest (slot, lane) holding it
    double lbest = sc[0];
#pragma unroll
    for (int s = 1; s < PER; ++s) lbest = fmax(lbest, sc[s]);
    double best = wave_max_f64(lbest);
    int bi = -1;
#pragma unroll
    for (int s = PER - 1; s >= 0; --s) {
      const unsigned long long m = __ballot(sc[s] == best);         // (every lane of every slot when the wave has nothing left: harmless)
      if (bi < 0 && m) bi = s * NT + wave * 64 + 63 - __clzll(m);
    }
    double px1 = 0, py1 = 0, px2 = 0, py2 = 0, pa = 0;
    {
      const int bs = bi / NT, bl = bi & 63;
#pragma unroll
      for (int s = 0; s < PER; ++s)
        if (s == bs) { px1 = x1[s]; py1 = y1[s]; px2 = x2[s]; py2 = y2[s]; pa = area[s]; }
      px1 = readlane_f64(px1, bl); py1 = readlane_f64(py1, bl); px2 = readlane_f64(px2, bl); py2 = readlane_f64(py2, bl); pa = readlane_f64(pa, bl);
    }
    bool stop = false;
    if constexpr (PRUNE) {
      if (wave == 0 && it > 0) {
        unsigned int part = (lane < kHistFine && lane > myfine) ? hf : 0u;
#pragma unroll
        for (int q = 0; q < 6; ++q) part += (lane * 6 + q > mybin) ? hc[q] : 0u;
        stop = wave_sum_u32(part) >= (unsigned int)g.top_k;
      }
    }
    if constexpr (WAVES > 1) {
      if (lane == 0) {
        Slot& sl = s_slot[par][wave];
        sl.best = best; sl.x1 = px1; sl.y1 = py1; sl.x2 = px2; sl.y2 = py2; sl.area = pa; sl.bi = bi;
        if (PRUNE && stop) s_stop[par] = 1;
      }
      __syncthreads();
      if constexpr (PRUNE) stop = s_stop[par] != 0;
      int ws = 0;
      best = s_slot[par][0].best;
#pragma unroll
      for (int w = 1; w < WAVES; ++w) {
        const double ob = s_slot[par][w].best;
        const int oi = s_slot[par][w].bi, ci = s_slot[par][ws].bi;
        if (ob > best || (ob == best && oi > ci)) { best = ob; ws = w; }
      }
      const Slot& sl = s_slot[par][ws];
      px1 = sl.x1; py1 = sl.y1; px2 = sl.x2; py2 = sl.y2; pa = sl.area; bi = sl.bi;
    }
    if (stop) break;                                      // every later pick of this class is <= the last one: below the image cut for good
    if (!(best > kGone)) break;                           // everything suppressed (hard NMS)
    if (tid == 0) {
      double* o = out + (long)picked * 5;
      o[0] = px1; o[1] = py1; o[2] = px2; o[3] = py2; o[4] = best;
      if (g.pick_index) g.pick_index[((long)b * (g.C - 1) + (cls - 1)) * g.N + picked] = bi;
    }
    ++picked;
    if constexpr (PRUNE) {
      if (wave == 0) {
        mybin = score_bin(best, myfine);
        unsigned int* hist2 = hist + kHistBins;
        if (lane == 0) {
          __hip_atomic_fetch_add(hist + mybin, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_fetch_add(hist2 + mybin * kHistFine + myfine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int q = 0; q < 6; ++q) hc[q] = __hip_atomic_load(hist + lane * 6 + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hf = lane < kHistFine ? __hip_atomic_load(hist2 + mybin * kHistFine + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      }
    }
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      if (s * NT + tid == bi) { sc[s] = kGone; continue; }
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

// One workgroup per image.  The picks of the image's classes are ONE list of `total` entries (class-major, pick order = the order
// of the output).  total <= kTopkKeys (always, once relnet_class_nms_topk has pruned the lists): keys and class ids are fetched once
// into LDS, the max_per_image-th largest key comes from an 8 x 8-bit radix select whose digit is found by one wavefront (suffix sums
// over the 256 bins, 4 per lane), and the output positions are an exclusive block scan of the keep flags -- five global round trips
// in all.  (Rounds 1-3 walked the class lists with one thread per class and scanned the bins with one thread: 70-130 us of
// dependent loads per call, as long as the class NMS itself.)  Longer lists take the general path below it.
__global__ __launch_bounds__(1024) void image_topk_kernel(ImgTopkArgs g) {
  __shared__ __attribute__((aligned(16))) unsigned int hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_total, s_out;
  __shared__ int s_off[129];                  // exclusive prefix of the class counts
  __shared__ int s_wsum[16];
  __shared__ unsigned long long s_keys[kTopkKeys];
  __shared__ unsigned char s_cls[kTopkKeys];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int* cnt = g.counts + (long)b * g.NC;
  const double* dets = g.dets + (long)b * g.NC * g.N * 5;
  if (tid < g.NC) s_off[tid + 1] = cnt[tid];
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    s_off[0] = 0;
    for (int c = 1; c <= g.NC; ++c) { t += s_off[c]; s_off[c] = t; }
    s_total = t; s_out = 0;
  }
  __syncthreads();
  const int total = s_total;
  auto class_of = [&](int i) {                // last class whose offset is <= i
    int lo = 0, hi = g.NC;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_off[mid] <= i) lo = mid; else hi = mid; }
    return lo;
  };
  if (total <= kTopkKeys) {
    for (int i = tid; i < total; i += 1024) {
      const int c = class_of(i);
      s_cls[i] = (unsigned char)c;
      s_keys[i] = dkey(dets[((long)c * g.N + (i - s_off[c])) * 5 + 4]);
    }
    if (tid == 0) { s_prefix = 0ull; s_remaining = g.max_per_image; }
    __syncthreads();
    unsigned long long kth = 0ull;
    if (total > g.max_per_image) {
      for (int pass = 7; pass >= 0; --pass) {
        if (tid < 256) hist[tid] = 0;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        const int shift = pass * 8;
        for (int i = tid; i < total; i += 1024) {
          const unsigned long long key = s_keys[i];
          if (pass == 7 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(key >> shift) & 0xff], 1u);
        }
        __syncthreads();
        if (wave == 0) {                      // digit = the highest bin d whose count from the top reaches `remaining` (0 if none)
          const unsigned int rem = (unsigned int)s_remaining;
          const uint4 h = ((const uint4*)hist)[lane];
          const unsigned int tot = h.x + h.y + h.z + h.w;
          unsigned int suf = tot;             // inclusive suffix sum over the lanes: keys in bins >= 4 lane
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) { const unsigned int v = __shfl_down(suf, o); if (lane + o < 64) suf += v; }
          const unsigned long long m = __ballot(suf >= rem);
          const int L = m ? 63 - __clzll(m) : 0;
          if (lane == L) {
            unsigned int r = rem - (suf - tot);       // still to find inside this lane's four bins
            int d = 4 * L;
            if (h.w >= r) d += 3;
            else { r -= h.w; if (h.z >= r) d += 2; else { r -= h.z; if (h.y >= r) d += 1; else r -= h.y; } }
            s_remaining = (int)r;
            s_prefix = prefix | ((unsigned long long)d << shift);
          }
        }
        __syncthreads();
      }
      kth = s_prefix;
    }
    if (tid == 0) {
      g.total[b] = total;
      double th = -INFINITY;
      if (total > g.max_per_image) {
        const unsigned long long u = (kth >> 63) ? (kth & 0x7fffffffffffffffull) : ~kth;
        th = __longlong_as_double((long long)u);
      }
      g.thresh[b] = th;
    }
    // keep score >= threshold, class-major / pick order (tester.py:273-277): thread t owns entries [t E, (t + 1) E)
    const int E = (total + 1023) / 1024;
    const int i0 = tid * E, i1 = min(i0 + E, total);
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += (s_keys[i] >= kth) ? 1 : 0;
    int inc = mine;                           // inclusive scan: wave, then the 16 wave totals
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o); if (lane >= o) inc += v; }
    if (lane == 63) s_wsum[wave] = inc;
    __syncthreads();
    int base = inc - mine;
    for (int w = 0; w < wave; ++w) base += s_wsum[w];
    if (tid == 1023) g.out_count[b] = (base + mine) < g.max_out ? (base + mine) : g.max_out;
    for (int i = i0; i < i1; ++i) {
      if (s_keys[i] < kth) continue;
      if (base < g.max_out) {
        const int c = s_cls[i];
        const double* d = dets + ((long)c * g.N + (i - s_off[c])) * 5;
        float* o = g.out + ((long)b * g.max_out + base) * 6;
        o[0] = (float)(c + 1); o[1] = (float)d[4]; o[2] = (float)d[0]; o[3] = (float)d[1]; o[4] = (float)d[2]; o[5] = (float)d[3];
      }
      ++base;
    }
    return;
  }
  // ---- general path: the keys are re-read from global memory in every pass
  auto key_of = [&](int i) {
    const int c = class_of(i);
    return dkey(dets[((long)c * g.N + (i - s_off[c])) * 5 + 4]);
  };
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
        const unsigned long long key = key_of(i);
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
  if (tid == 0) {
    g.total[b] = total;
    double th = -INFINITY;
    if (total > g.max_per_image) {
      const unsigned long long u = (kth >> 63) ? (kth & 0x7fffffffffffffffull) : ~kth;
      th = __longlong_as_double((long long)u);
    }
    g.thresh[b] = th;
  }
  __syncthreads();                            // (s_off is rewritten below: every class_of() above is done)
  // per-class kept counts -> offsets
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

// One candidate per thread, the smallest workgroup that holds N of them (300 rois: 5 waves; FPN's 1000 + 4: 16).  (The one-wave form --
// N / 64 candidates per lane, half the instructions per pick -- was kept for launches of >= 1536 (image, class) pairs at first: faster
// in isolation at 54 images, 206 vs 242 us, but inside the step the multi-wave form wins or ties at every batch size, 19.19 vs
// 19.22-19.39 ms at 54 images: the step ends with this kernel, and its tail is the deepest class's serial picks.)
template <bool PRUNE>
static void launch_class_nms(const ClsNmsArgs& g, int B, hipStream_t s) {
  dim3 grid(g.C - 1, B);
  if (g.N <= 64) class_nms_kernel<1, 1, PRUNE><<<grid, 64, 0, s>>>(g);
  else if (g.N <= 128) class_nms_kernel<2, 1, PRUNE><<<grid, 128, 0, s>>>(g);
  else if (g.N <= 320) class_nms_kernel<5, 1, PRUNE><<<grid, 320, 0, s>>>(g);
  else if (g.N <= 512) class_nms_kernel<8, 1, PRUNE><<<grid, 512, 0, s>>>(g);
  else class_nms_kernel<16, 1, PRUNE><<<grid, 1024, 0, s>>>(g);
}

extern "C" int relnet_class_nms_ex(const float* cls_prob, const double* scores64, const double* boxes, double* dets,
                                   int* counts, int* pick_index, int B, int N, int C, float score_thresh,
                                   double nms_param, int soft, int max_picks, void* stream) {
  RELNET_REQUIRE((cls_prob || scores64) && boxes && dets && counts, "relnet_class_nms: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && N <= 1024 && C > 1, "relnet_class_nms: need 0 < N <= 1024 (N=%d)", N);
  RELNET_REQUIRE(!scores64 || C == 2, "relnet_class_nms: float64 scores are one foreground class (C == 2), got C=%d", C);
  ClsNmsArgs g{cls_prob, boxes, dets, counts, N, C, score_thresh, nms_param, soft, max_picks > 0 ? max_picks : N,
               scores64, pick_index, nullptr, 0};
  launch_class_nms<false>(g, B, (hipStream_t)stream);
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
// ZEROED by the caller before every call.  N <= 1024.
extern "C" int relnet_class_nms_hist_bins(void) { return kHistWords; }
extern "C" int relnet_class_nms_topk(const float* cls_prob, const double* boxes, double* dets, int* counts, void* hist, int B, int N,
                                     int C, float score_thresh, double nms_param, int soft, int max_picks, int top_k, void* stream) {
  RELNET_REQUIRE(cls_prob && boxes && dets && counts && hist, "relnet_class_nms_topk: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && N <= 1024 && C > 1 && top_k > 0, "relnet_class_nms_topk: need 0 < N <= 1024, top_k > 0 (N=%d top_k=%d)", N, top_k);
  ClsNmsArgs g{cls_prob, boxes, dets, counts, N, C, score_thresh, nms_param, soft, max_picks > 0 ? max_picks : N, nullptr, nullptr,
               (unsigned int*)hist, top_k};
  launch_class_nms<true>(g, B, (hipStream_t)stream);
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
