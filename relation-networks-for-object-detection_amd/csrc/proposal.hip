// RPN proposal path, entirely on device (reference: relation_rcnn/operator_py/proposal.py:51-168
// with lib/bbox/bbox_transform.py:103-140,45-60 and lib/nms/nms_kernel.cu:24-144; the
// reference round-trips through host numpy four times per image here).
//
//   proposal_decode_kernel  anchors (y,x,a order) + deltas -> clipped boxes; decode in
//                           float64 exactly like the numpy reference, cast to float32.
//   topk_sort_kernel        descending top-K of the scores: 48-bit radix select of the
//                           K-th key, compaction, bitonic sort in LDS (one workgroup / image).
//   nms_mask_kernel         64x64 IoU tiles -> uint64 suppression bitmask (upper triangle).
//   nms_scan_kernel         greedy scan on device (the reference copies 4.5 MB of mask to
//                           the host and scans serially), early exit at post_nms_top_n,
//                           gathers the rois.
// Tie rule (the reference's unstable argsort leaves it open): equal scores are ordered by
// descending original index, i.e. `argsort(kind='stable')[::-1]`.
#include "common.h"

namespace relnet {

// ---------------------------------------------------------------------------------------
struct DecodeArgs {
  const float* cls_prob;  long cs_b, cs_c, cs_h, cs_w;   // [B, 2A, H, W] element strides
  const float* deltas;    long ds_b, ds_c, ds_h, ds_w;   // [B, 4A, H, W]
  const float* im_info;   // [B, 3] (h, w, scale)
  const double* anchors;  // [A, 4] base anchors
  float* boxes;           // [B, n, 4] n = h*w*A (cropped grid)
  float* scores;          // [B, n]    (-inf when filtered by min_size)
  int A, h, w, feat_stride, min_size;
  int softmax_pairs;      // 1: cls_prob holds raw rpn_cls_score, fg prob = softmax over {bg=a, fg=A+a}
};

#pragma clang fp contract(off)
__global__ __launch_bounds__(256) void proposal_decode_kernel(DecodeArgs g) {
  const int n = g.h * g.w * g.A;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (idx >= n) return;
  const int a = idx % g.A, x = (idx / g.A) % g.w, y = idx / (g.A * g.w);
  const float* info = g.im_info + b * 3;
  float score = g.cls_prob[b * g.cs_b + (long)(g.A + a) * g.cs_c + y * g.cs_h + x * g.cs_w];
  if (g.softmax_pairs) {   // SoftmaxActivation(mode='channel') on the (0,2,-1,0) reshape, SYM_REL:218-221
    const float bg = g.cls_prob[b * g.cs_b + (long)a * g.cs_c + y * g.cs_h + x * g.cs_w];
    const float m = fmaxf(bg, score);
    const float eb = expf(bg - m), ef = expf(score - m);
    score = ef / (eb + ef);
  }
  const float* dp = g.deltas + b * g.ds_b + (long)(4 * a) * g.ds_c + y * g.ds_h + x * g.ds_w;
  const float dx = dp[0], dy = dp[g.ds_c], dw = dp[2 * g.ds_c], dh = dp[3 * g.ds_c];
  const double sx = (double)(x * g.feat_stride), sy = (double)(y * g.feat_stride);
  const double x1 = g.anchors[a * 4 + 0] + sx, y1 = g.anchors[a * 4 + 1] + sy;
  const double x2 = g.anchors[a * 4 + 2] + sx, y2 = g.anchors[a * 4 + 3] + sy;
  // bbox_transform.py:114-138 in float64; np.exp on the float32 deltas = fp32 exp
  const double bw = x2 - x1 + 1.0, bh = y2 - y1 + 1.0;
  const double cx = x1 + 0.5 * (bw - 1.0), cy = y1 + 0.5 * (bh - 1.0);
  const double pcx = (double)dx * bw + cx, pcy = (double)dy * bh + cy;
  const double pw = (double)(float)exp((double)dw) * bw;
  const double ph = (double)(float)exp((double)dh) * bh;
  double ox1 = pcx - 0.5 * (pw - 1.0), oy1 = pcy - 0.5 * (ph - 1.0);
  double ox2 = pcx + 0.5 * (pw - 1.0), oy2 = pcy + 0.5 * (ph - 1.0);
  const double mx = (double)info[1] - 1.0, my = (double)info[0] - 1.0;   // clip_boxes :45-60
  ox1 = fmax(fmin(ox1, mx), 0.0); oy1 = fmax(fmin(oy1, my), 0.0);
  ox2 = fmax(fmin(ox2, mx), 0.0); oy2 = fmax(fmin(oy2, my), 0.0);
  const double ms = (double)g.min_size * (double)info[2];                 // proposal.py:133-135
  const bool ok = (ox2 - ox1 + 1.0 >= ms) && (oy2 - oy1 + 1.0 >= ms);
  *(float4*)(g.boxes + ((long)b * n + idx) * 4) = make_float4((float)ox1, (float)oy1, (float)ox2, (float)oy2);
  g.scores[(long)b * n + idx] = ok ? score : -INFINITY;
}

// order-preserving float -> uint32 (larger float => larger key); -inf smallest finite-ish
__device__ __forceinline__ unsigned int fkey(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------------------------------
// top-K + sort.  Composite key = (fkey(score) << 16) | idx  (idx < 65536), descending.
// ---------------------------------------------------------------------------------------
constexpr int kSortCap = 8192;       // LDS bitonic capacity (64 KiB of uint64)
constexpr int kSortThreads = 1024;

struct TopkArgs {
  const float* scores;   // [B, n]
  const float* boxes;    // [B, n, 4]
  float* out_boxes;      // [B, K, 5]  x1,y1,x2,y2,score  (sorted, the `det` of proposal.py:149)
  int* out_index;        // [B, K] original anchor index (or -1)
  int* out_count;        // [B] number of valid (finite score) entries <= K
  int n, K;
};

// One workgroup per image; thread t keeps the score keys of anchors t, t + 1024, ... (PER of them: n <= 1024 PER) in REGISTERS for
// the whole kernel -- the scores are read from memory once, in one batch of independent loads (rounds 1-3 re-read them in each
// of the seven passes, one dependent L2 round trip per element: 240-440 us per call, all latency).
//   radix select   the K-th largest 48-bit composite key in six 8-bit passes.  Histogram updates are aggregated per wavefront for the
//                  digit of its first matching lane (objectness scores share their top bits: a plain LDS atomic would serialise
//                  64 ways), the rest go in as plain atomics; the digit is found by one wavefront (suffix sums over 4 bins per lane);
//   compaction     one LDS counter update per wavefront (ballot + prefix popcount);
//   bitonic sort   all threads busy in every stage (thread = pair index, not element index).
template <int PER>
__global__ __launch_bounds__(kSortThreads) void topk_sort_kernel(TopkArgs g) {
  __shared__ unsigned long long keys[kSortCap];
  __shared__ __attribute__((aligned(16))) unsigned int hist[256];
  __shared__ unsigned long long s_prefix;
  __shared__ int s_remaining, s_count, s_valid;
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* sc = g.scores + (long)b * g.n;
  const int K = g.K < g.n ? g.K : g.n;
  unsigned int fk[PER];
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const int i = s * kSortThreads + tid;
    fk[s] = i < g.n ? fkey(sc[i]) : 0u;
  }
  // ---- radix select: find the K-th largest composite key (fkey << 16 | index) --------------------
  if (tid == 0) { s_prefix = 0ull; s_remaining = K; s_count = 0; s_valid = 0; }
  for (int i = tid; i < kSortCap; i += kSortThreads) keys[i] = 0ull;
  __syncthreads();
  for (int pass = 5; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    const unsigned long long prefix = s_prefix;
    const int shift = pass * 8;
#pragma unroll
    for (int s = 0; s < PER; ++s) {
      const int i = s * kSortThreads + tid;
      const unsigned long long ck = ((unsigned long long)fk[s] << 16) | (unsigned int)i;
      const bool match = i < g.n && (pass == 5 || (ck >> (shift + 8)) == (prefix >> (shift + 8)));
      const int d = (int)((ck >> shift) & 0xff);
      const unsigned long long mm = __ballot(match);
      if (mm) {
        const int lead = __builtin_ctzll(mm);
        const int dl = __builtin_amdgcn_readlane(d, lead);
        const unsigned long long peers = __ballot(match && d == dl);
        if (lane == lead) atomicAdd(&hist[dl], (unsigned int)__popcll(peers));
        else if (match && d != dl) atomicAdd(&hist[d], 1u);
      }
    }
    __syncthreads();
    if (wave == 0) {                      // digit = the highest bin whose count from the top reaches `remaining` (0 if none)
      const unsigned int rem = (unsigned int)s_remaining;
      const uint4 h = ((const uint4*)hist)[lane];
      const unsigned int tot = h.x + h.y + h.z + h.w;
      unsigned int suf = tot;             // inclusive suffix sum over the lanes: keys in bins >= 4 lane
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const unsigned int v = __shfl_down(suf, o); if (lane + o < 64) suf += v; }
      const unsigned long long m = __ballot(suf >= rem);
      const int L = m ? 63 - __clzll(m) : 0;
      if (lane == L) {
        unsigned int r = rem - (suf - tot);
        int d = 4 * L;
        if (h.w >= r) d += 3;
        else { r -= h.w; if (h.z >= r) d += 2; else { r -= h.z; if (h.y >= r) d += 1; else r -= h.y; } }
        s_remaining = (int)r;
        s_prefix = prefix | ((unsigned long long)d << shift);
      }
    }
    __syncthreads();
  }
  const unsigned long long kth = s_prefix;      // exactly K composite keys are >= kth
#pragma unroll
  for (int s = 0; s < PER; ++s) {
    const int i = s * kSortThreads + tid;
    const unsigned long long ck = ((unsigned long long)fk[s] << 16) | (unsigned int)i;
    const bool take = i < g.n && ck >= kth;
    const unsigned long long mm = __ballot(take);
    if (mm) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(mm));
      base = __builtin_amdgcn_readfirstlane(base);
      if (take) keys[base + __popcll(mm & ((1ull << lane) - 1ull))] = ck;
    }
  }
  __syncthreads();
  // ---- bitonic sort, descending, over the next power of two >= K: thread = pair (i, i | j) ---------------------
  int np2 = 1;
  while (np2 < K) np2 <<= 1;
  for (int k = 2; k <= np2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < (np2 >> 1); p += kSortThreads) {
        const int i = ((p & ~(j - 1)) << 1) | (p & (j - 1)), ixj = i | j;
        const unsigned long long a = keys[i], c = keys[ixj];
        const bool desc = (i & k) == 0;
        if (desc ? (a < c) : (a > c)) { keys[i] = c; keys[ixj] = a; }
      }
      __syncthreads();
    }
  }
  // ---- gather ------------------------------------------------------------------------
  int valid = 0;
  for (int i = tid; i < K; i += kSortThreads) {
    const unsigned long long ck = keys[i];
    const int idx = (int)(ck & 0xffffu);
    const float s = sc[idx];
    const float4 bx = *(const float4*)(g.boxes + ((long)b * g.n + idx) * 4);
    float* o = g.out_boxes + ((long)b * g.K + i) * 5;
    o[0] = bx.x; o[1] = bx.y; o[2] = bx.z; o[3] = bx.w; o[4] = s;
    g.out_index[(long)b * g.K + i] = idx;
    valid += (s > -INFINITY) ? 1 : 0;
  }
  // count of finite-score entries (filtered boxes carry -inf and sort last)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) valid += __shfl_xor(valid, o);
  if (lane == 0 && valid) atomicAdd(&s_valid, valid);
  __syncthreads();
  if (tid == 0) g.out_count[b] = s_valid;
}

// ---------------------------------------------------------------------------------------
// NMS bitmask: box rows are [x1,y1,x2,y2,score] (stride 5) sorted by score.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float dev_iou(const float* a, const float* b) {
  // nms_kernel.cu:24-32, one rounding per operation (no FMA contraction)
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left + 1.f, 0.f), height = fmaxf(bottom - top + 1.f, 0.f);
  const float inter = width * height;
  const float sa = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
  const float sb = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
  return inter / (sa + sb - inter);
}

struct MaskArgs {
  const float* boxes;            // [B, n_stride, 5]
  const int* counts;             // [B] valid boxes per image (or nullptr -> n)
  unsigned long long* mask;      // [B, n, col_blocks]
  int n, n_stride, col_blocks;
  float thresh;
};

__global__ __launch_bounds__(64) void nms_mask_kernel(MaskArgs g) {
  const int row_blk = blockIdx.y, col_blk = blockIdx.x, b = blockIdx.z;
  if (col_blk < row_blk) return;                       // upper triangle only
  const int n = g.counts ? min(g.counts[b], g.n) : g.n;
  if (row_blk * 64 >= n) return;
  const float* bx = g.boxes + (long)b * g.n_stride * 5;
  const int row_size = min(n - row_blk * 64, 64), col_size = min(n - col_blk * 64, 64);
  __shared__ float blk[64 * 5];
  const int t = threadIdx.x;
  if (t < col_size) {
#pragma unroll
    for (int c = 0; c < 4; ++c) blk[t * 5 + c] = bx[(long)(col_blk * 64 + t) * 5 + c];
  }
  __syncthreads();
  if (t < row_size) {
    const int cur = row_blk * 64 + t;
    float me[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) me[c] = bx[(long)cur * 5 + c];
    unsigned long long bits = 0ull;
    const int start = (row_blk == col_blk) ? t + 1 : 0;
    for (int i = start; i < col_size; ++i)
      if (dev_iou(me, blk + i * 5) > g.thresh) bits |= 1ull << i;
    g.mask[((long)b * g.n + cur) * g.col_blocks + col_blk] = bits;
  }
}
#pragma clang fp contract(fast)

struct ScanArgs {
  const unsigned long long* mask;   // [B, n, col_blocks] (only upper-triangle words valid)
  const float* boxes;               // [B, n_stride, 5]
  const int* counts;                // [B] or nullptr
  float* rois;                      // [B, post, 5] (batch index, x1,y1,x2,y2) or nullptr
  float* roi_scores;                // [B, post] or nullptr
  int* keep;                        // [B, max_keep] kept positions (ascending) or nullptr
  int* num_keep;                    // [B]
  int n, n_stride, col_blocks, post, max_keep, batch_index_base;
};

// One workgroup (4 waves) per image.  Per 64-box block: wave 0 resolves the intra-block
// order with the diagonal mask words (ctz + readlane), then all waves OR the kept boxes'
// mask rows into the LDS `removed` bitmap.
__global__ __launch_bounds__(256) void nms_scan_kernel(ScanArgs g) {
  __shared__ unsigned long long remv[1024];
  __shared__ unsigned long long s_keepmask;
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int n = g.counts ? min(g.counts[b], g.n) : g.n;
  const unsigned long long* mask = g.mask + (long)b * g.n * g.col_blocks;
  const int limit = g.post > 0 ? min(g.post, g.max_keep) : g.max_keep;
  for (int i = tid; i < g.col_blocks; i += 256) remv[i] = 0ull;
  if (tid == 0) s_total = 0;
  __syncthreads();
  const int nblk = (n + 63) / 64;
  for (int blk = 0; blk < nblk; ++blk) {
    if (wave == 0) {
      const int i = blk * 64 + lane;
      const unsigned long long diag = (i < n) ? mask[(long)i * g.col_blocks + blk] : 0ull;
      const int bsize = min(n - blk * 64, 64);
      unsigned long long cur = ~remv[blk];
      if (bsize < 64) cur &= (1ull << bsize) - 1ull;
      unsigned long long km = 0ull;
      int total = s_total;
      while (cur != 0ull && total < limit) {
        const int t0 = __builtin_ctzll(cur);
        km |= 1ull << t0;
        ++total;
        const unsigned long long d0 = __shfl(diag, t0);
        cur &= ~d0;
        cur &= ~(1ull << t0);
      }
      // record kept positions in order
      if ((km >> lane) & 1ull) {
        const int pos = s_total + __builtin_popcountll(km & ((1ull << lane) - 1ull));
        if (g.keep && pos < g.max_keep) g.keep[(long)b * g.max_keep + pos] = i;
        if (g.rois && pos < g.post) {
          const float* bx = g.boxes + ((long)b * g.n_stride + i) * 5;
          float* r = g.rois + ((long)b * g.post + pos) * 5;
          r[0] = (float)(g.batch_index_base + b); r[1] = bx[0]; r[2] = bx[1]; r[3] = bx[2]; r[4] = bx[3];
          if (g.roi_scores) g.roi_scores[(long)b * g.post + pos] = bx[4];
        }
      }
      if (lane == 0) { s_keepmask = km; s_total = total; }
    }
    __syncthreads();
    const unsigned long long km = s_keepmask;
    if (s_total >= limit) break;
    // OR mask rows of kept boxes into remv[j], j > blk
    unsigned long long rest = km;
    int ord = 0;
    while (rest != 0ull) {
      const int t0 = __builtin_ctzll(rest);
      rest &= rest - 1ull;
      if ((ord++ & 3) == wave) {
        const unsigned long long* row = mask + (long)(blk * 64 + t0) * g.col_blocks;
        for (int j = blk + 1 + lane; j < g.col_blocks; j += 64) {
          const unsigned long long w = row[j];
          if (w) atomicOr(&remv[j], w);
        }
      }
    }
    __syncthreads();
  }
  __syncthreads();
  const int total = s_total;
  if (tid == 0) g.num_keep[b] = total;
  // pad rois when fewer than `post` survive: the reference pads with a RANDOM choice of the
  // kept boxes (proposal.py:154-156); here deterministically keep[i mod total].
  if (g.rois && total < g.post && total > 0) {
    for (int pos = total + tid; pos < g.post; pos += 256) {
      const float* src = g.rois + ((long)b * g.post + (pos % total)) * 5;
      float* dst = g.rois + ((long)b * g.post + pos) * 5;
      for (int c = 0; c < 5; ++c) dst[c] = src[c];
      if (g.roi_scores) g.roi_scores[(long)b * g.post + pos] = g.roi_scores[(long)b * g.post + (pos % total)];
    }
  }
}


// ---------------------------------------------------------------------------------------
// Greedy NMS with a bounded keep list, fused: instead of the full n x n suppression bitmask
// (18 M IoUs + 4.5 MB of mask per image at n = 6000) each 64-candidate block is tested
// against the <= max_keep boxes kept so far and against itself, and the scan stops as soon as
// `post` boxes are kept -- the only rows proposal.py:151-153 uses.  Same IoU arithmetic and
// the same greedy order as nms_mask_kernel + nms_scan_kernel, hence the same keep list.
// ---------------------------------------------------------------------------------------
constexpr int kMaxKeep = 2048;

struct GreedyArgs {
  const float* boxes;      // [B, n_stride, 5] sorted by score
  const int* counts;       // [B] or nullptr
  float* rois;             // [B, post, 5] or nullptr
  float* roi_scores;       // [B, post] or nullptr
  int* keep;               // [B, post] kept positions or nullptr
  int* num_keep;           // [B]
  int n, n_stride, post, batch_index_base;
  float thresh;
};

constexpr int kGreedyWaves = 16;

#pragma clang fp contract(off)
__global__ __launch_bounds__(64 * kGreedyWaves) void nms_greedy_kernel(GreedyArgs g) {
  __shared__ float kb[kMaxKeep * 4];
  __shared__ unsigned long long s_sup[kGreedyWaves];
  __shared__ unsigned long long s_diag[64];
  __shared__ int s_total;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  const int n = g.counts ? min(g.counts[b], g.n) : g.n;
  const float* bx = g.boxes + (long)b * g.n_stride * 5;
  if (tid == 0) s_total = 0;
  __syncthreads();
  const int nblk = (n + 63) / 64;
  float cn[4] = {0.f, 0.f, 0.f, 0.f};                  // the NEXT block's boxes: requested one block ahead of their use
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < 4; ++k) cn[k] = bx[(long)lane * 5 + k];
  }
  for (int blk = 0; blk < nblk; ++blk) {
    const int i = blk * 64 + lane;
    const int bsize = min(n - blk * 64, 64);
    float c[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = cn[k]; cn[k] = 0.f; }
    if (i + 64 < n) {
#pragma unroll
      for (int k = 0; k < 4; ++k) cn[k] = bx[(long)(i + 64) * 5 + k];
    }
    const int total = s_total;
    if (tid < 64) s_diag[tid] = 0ull;
    // ---- phase 1: candidates vs boxes already kept (the waves split the kept list) ---------
    bool sup = false;
    for (int k = wave; k < total; k += kGreedyWaves) sup = sup || (dev_iou(kb + 4 * k, c) > g.thresh);
    const unsigned long long ball = __ballot(sup);
    if (lane == 0) s_sup[wave] = ball;
    __syncthreads();
    // ---- phase 2: order inside the block: bits j > lane with IoU(lane, j) > thresh; each wave
    // covers 4 of the 64 rotation offsets -----------------------------------------------------
    unsigned long long part = 0ull;
#pragma unroll
    for (int jj = 0; jj < 64 / kGreedyWaves; ++jj) {
      const int off = wave * (64 / kGreedyWaves) + jj;
      const int src = (lane + off) & 63;
      float o[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] = __shfl(c[k], src);
      if (off != 0 && src > lane && src < bsize && dev_iou(c, o) > g.thresh) part |= 1ull << src;
    }
    if (part) atomicOr(&s_diag[lane], part);
    __syncthreads();
    if (wave == 0) {
      unsigned long long alive = 0ull;
#pragma unroll
      for (int w = 0; w < kGreedyWaves; ++w) alive |= s_sup[w];
      alive = ~alive;
      if (bsize < 64) alive &= (1ull << bsize) - 1ull;
      const unsigned long long diag = s_diag[lane];
      const int dlo = (int)(unsigned int)diag, dhi = (int)(unsigned int)(diag >> 32);
      // the walk over the block is wave-uniform: kept on the scalar unit (row t0 of the in-block matrix through v_readlane)
      unsigned long long cur = ((unsigned long long)(unsigned int)__builtin_amdgcn_readfirstlane((int)(alive >> 32)) << 32) |
                               (unsigned int)__builtin_amdgcn_readfirstlane((int)(unsigned int)alive);
      unsigned long long km = 0ull;
      int tot = __builtin_amdgcn_readfirstlane(total);
      while (cur != 0ull && tot < g.post) {
        const int t0 = __builtin_ctzll(cur);
        km |= 1ull << t0;
        ++tot;
        const unsigned long long d0 = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane(dhi, t0) << 32) |
                                      (unsigned int)__builtin_amdgcn_readlane(dlo, t0);
        cur &= ~d0;
        cur &= ~(1ull << t0);
      }
      if ((km >> lane) & 1ull) {
        const int pos = total + __builtin_popcountll(km & ((1ull << lane) - 1ull));
#pragma unroll
        for (int k = 0; k < 4; ++k) kb[4 * pos + k] = c[k];
        if (g.keep) g.keep[(long)b * g.post + pos] = i;
        if (g.rois) {
          float* r = g.rois + ((long)b * g.post + pos) * 5;
          r[0] = (float)(g.batch_index_base + b); r[1] = c[0]; r[2] = c[1]; r[3] = c[2]; r[4] = c[3];
          if (g.roi_scores) g.roi_scores[(long)b * g.post + pos] = bx[(long)i * 5 + 4];
        }
      }
      if (lane == 0) s_total = tot;
    }
    __syncthreads();
    if (s_total >= g.post) break;
  }
  const int total = s_total;
  if (tid == 0) g.num_keep[b] = total;
  if (g.rois && total < g.post && total > 0) {         // deterministic padding (see nms_scan_kernel)
    for (int pos = total + tid; pos < g.post; pos += 64 * kGreedyWaves) {
      const float* src = g.rois + ((long)b * g.post + (pos % total)) * 5;
      float* dst = g.rois + ((long)b * g.post + pos) * 5;
      for (int k = 0; k < 5; ++k) dst[k] = src[k];
      if (g.roi_scores) g.roi_scores[(long)b * g.post + pos] = g.roi_scores[(long)b * g.post + (pos % total)];
    }
  }
}
#pragma clang fp contract(fast)

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_proposal_decode(const float* cls_prob, const long* cls_strides4,
                                      const float* deltas, const long* delta_strides4,
                                      const float* im_info, const double* base_anchors, float* boxes,
                                      float* scores, int B, int A, int h, int w, int feat_stride,
                                      int min_size, int softmax_pairs, void* stream) {
  RELNET_REQUIRE(cls_prob && deltas && im_info && base_anchors && boxes && scores, "relnet_proposal_decode: null operand");
  RELNET_REQUIRE(B > 0 && A > 0 && h > 0 && w > 0 && (long)h * w * A < 65536, "relnet_proposal_decode: bad grid B=%d A=%d h=%d w=%d (h*w*A must be < 65536)", B, A, h, w);
  DecodeArgs g;
  g.cls_prob = cls_prob; g.cs_b = cls_strides4[0]; g.cs_c = cls_strides4[1]; g.cs_h = cls_strides4[2]; g.cs_w = cls_strides4[3];
  g.deltas = deltas; g.ds_b = delta_strides4[0]; g.ds_c = delta_strides4[1]; g.ds_h = delta_strides4[2]; g.ds_w = delta_strides4[3];
  g.im_info = im_info; g.anchors = base_anchors; g.boxes = boxes; g.scores = scores;
  g.A = A; g.h = h; g.w = w; g.feat_stride = feat_stride; g.min_size = min_size; g.softmax_pairs = softmax_pairs;
  dim3 grid((h * w * A + 255) / 256, B);
  proposal_decode_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_proposal_decode");
}

extern "C" int relnet_topk_sort(const float* scores, const float* boxes, float* out_boxes5,
                                int* out_index, int* out_count, int B, int n, int K, void* stream) {
  RELNET_REQUIRE(scores && boxes && out_boxes5 && out_index && out_count, "relnet_topk_sort: null operand");
  RELNET_REQUIRE(B > 0 && n > 0 && n < 65536 && K > 0 && K <= kSortCap, "relnet_topk_sort: need 0 < n < 65536 and 0 < K <= %d (n=%d K=%d)", kSortCap, n, K);
  TopkArgs g{scores, boxes, out_boxes5, out_index, out_count, n, K};
  if (n <= 32 * kSortThreads) topk_sort_kernel<32><<<B, kSortThreads, 0, (hipStream_t)stream>>>(g);
  else topk_sort_kernel<64><<<B, kSortThreads, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_topk_sort");
}

extern "C" int relnet_nms_mask(const float* boxes5, const int* counts, unsigned long long* mask,
                               int B, int n, int n_stride, float thresh, void* stream) {
  RELNET_REQUIRE(boxes5 && mask && B > 0 && n > 0 && n_stride >= n, "relnet_nms_mask: bad arguments");
  const int cb = (n + 63) / 64;
  MaskArgs g{boxes5, counts, mask, n, n_stride, cb, thresh};
  dim3 grid(cb, cb, B);
  nms_mask_kernel<<<grid, 64, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_nms_mask");
}

extern "C" int relnet_nms_scan(const unsigned long long* mask, const float* boxes5, const int* counts,
                               float* rois, float* roi_scores, int* keep, int* num_keep, int B,
                               int n, int n_stride, int post, int max_keep, int batch_index_base,
                               void* stream) {
  RELNET_REQUIRE(mask && boxes5 && num_keep && B > 0 && n > 0, "relnet_nms_scan: bad arguments");
  const int cb = (n + 63) / 64;
  RELNET_REQUIRE(cb <= 1024, "relnet_nms_scan: n=%d exceeds 65536 boxes", n);
  RELNET_REQUIRE(max_keep > 0 && (!rois || post > 0), "relnet_nms_scan: max_keep/post must be positive");
  ScanArgs g{mask, boxes5, counts, rois, roi_scores, keep, num_keep, n, n_stride, cb, post, max_keep, batch_index_base};
  nms_scan_kernel<<<B, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_nms_scan");
}

// ---------------------------------------------------------------------------------------
// `_nms`: the reference's C prototype (lib/nms/gpu_nms.hpp:1-2), host pointers in and out,
// boxes pre-sorted by score, rows [x1,y1,x2,y2,score].  Drop-in for gpu_nms.pyx:31.
// Unlike the reference (errors only printed, nms_kernel.cu:12-19) a HIP error leaves
// *num_out = -1 and is retrievable through relnet_last_error().
// ---------------------------------------------------------------------------------------
extern "C" void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
                     int boxes_dim, float nms_overlap_thresh, int device_id) {
  *num_out = -1;
  if (boxes_num <= 0) { *num_out = 0; return; }
  if (boxes_dim != 5) { set_error("_nms: boxes_dim must be 5 (got %d)", boxes_dim); return; }
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || (cur != device_id && hipSetDevice(device_id) != hipSuccess)) {
    set_error("_nms: cannot select device %d", device_id); return;
  }
  const int cb = (boxes_num + 63) / 64;
  float* d_boxes = nullptr; unsigned long long* d_mask = nullptr; int* d_keep = nullptr; int* d_num = nullptr;
  hipError_t e = hipMalloc(&d_boxes, sizeof(float) * 5 * boxes_num);
  if (e == hipSuccess) e = hipMalloc(&d_mask, sizeof(unsigned long long) * (size_t)boxes_num * cb);
  if (e == hipSuccess) e = hipMalloc(&d_keep, sizeof(int) * boxes_num);
  if (e == hipSuccess) e = hipMalloc(&d_num, sizeof(int));
  if (e == hipSuccess) e = hipMemcpy(d_boxes, boxes_host, sizeof(float) * 5 * boxes_num, hipMemcpyHostToDevice);
  if (e == hipSuccess && relnet_nms_mask(d_boxes, nullptr, d_mask, 1, boxes_num, boxes_num, nms_overlap_thresh, nullptr) == 0 &&
      relnet_nms_scan(d_mask, d_boxes, nullptr, nullptr, nullptr, d_keep, d_num, 1, boxes_num, boxes_num, 0, boxes_num, 0, nullptr) == 0) {
    int num = 0;
    e = hipMemcpy(&num, d_num, sizeof(int), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(keep_out, d_keep, sizeof(int) * num, hipMemcpyDeviceToHost);
    if (e == hipSuccess) *num_out = num;
  }
  if (e != hipSuccess) set_error("_nms: %s", hipGetErrorString(e));
  hipFree(d_boxes); hipFree(d_mask); hipFree(d_keep); hipFree(d_num);
}

// Fused greedy NMS keeping the first `post` (<= 2048) boxes: rois [B,post,5], roi_scores [B,post],
// keep [B,post] positions into the sorted boxes, num_keep [B].  Same result as relnet_nms_mask +
// relnet_nms_scan(post) without materialising the mask.
extern "C" int relnet_nms_greedy(const float* boxes5, const int* counts, float* rois, float* roi_scores,
                                 int* keep, int* num_keep, int B, int n, int n_stride, int post,
                                 float thresh, int batch_index_base, void* stream) {
  RELNET_REQUIRE(boxes5 && num_keep && B > 0 && n > 0 && n_stride >= n, "relnet_nms_greedy: bad arguments");
  RELNET_REQUIRE(post > 0 && post <= kMaxKeep, "relnet_nms_greedy: need 0 < post <= %d (post=%d)", kMaxKeep, post);
  GreedyArgs g{boxes5, counts, rois, roi_scores, keep, num_keep, n, n_stride, post, batch_index_base, thresh};
  nms_greedy_kernel<<<B, 64 * kGreedyWaves, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_nms_greedy");
}
