// FPN configuration.
//
//  * relnet_fpn_roi_dispatch : ROI -> pyramid level assignment and the stable per-level regrouping the
//      reference does on the host (relation_rcnn/core/rcnn.py:53-74, cfg.network.ROIDispatch):
//        feat_id = clip(floor(2 + log2(sqrt(w*h) / 224)), 0, 3),  w = x2-x1+1, h = y2-y1+1
//        rois_l  = rois[feat_id == l]   (original order inside a level), then concat l = 0..3
//      (symbols/resnet_v1_101_rcnn_fpn_..._learn_nms.py:1108-1121 concatenates pooled features and rois in
//      that level order).  The arithmetic is float32 like numpy's on float32 boxes, with log2 evaluated
//      correctly rounded (fp64 log2 rounded once).
//      relnet_fpn_roi_dispatch_ex(pad_empty = 1) also appends the all-zero dummy roi of an EMPTY level (rcnn.py:61-71)
//      into a fixed [B, N + 4] row buffer and reports the real row count per image (`n_rows`); rows past it are
//      padding that the relation kernels skip as keys (`key_count`) and the post-processing kernels drop (`n_valid`).
//  * relnet_upsample2x_add : mx.symbol.UpSampling(scale=2, sample_type='nearest') + ElementWiseSum of the
//      top-down pathway (symbols/...fpn...:817-829), in place on the lateral map.
#include "common.h"

namespace relnet {

enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

struct DispatchArgs {
  const float* rois; int box_stride, box_off;      // [B, N, box_stride], xyxy at box_off
  float* rois_out;                                  // [B, n_out, 5] level-sorted, column 0 = image index + base
  int* level_out;                                   // [B, n_out] level of each sorted row
  int* perm;                                        // [B, n_out] original index of each sorted row (-1: dummy / unused row)
  int* counts;                                      // [B, 4] input rois per level (dummies not counted)
  int N, batch_index_base;
  const int* n_valid;                               // [B] or nullptr: only the first n_valid[b] input rows are rois
  int pad_empty, n_out;                             // pad_empty: one all-zero roi for every level without a roi (rcnn.py:61-71)
  int* n_rows;                                      // [B] or nullptr: rows of this image that are real (rois + dummies)
};

#pragma clang fp contract(off)
__device__ __forceinline__ int fpn_level(float x1, float y1, float x2, float y2) {
  const float w = x2 - x1 + 1.f, h = y2 - y1 + 1.f;
  const float s = sqrtf(w * h) / 224.f;                       // correctly rounded sqrt / divide
  const float l2 = (float)log2((double)s);                    // correctly rounded fp32 log2
  const float v = floorf(2.f + l2);
  return (int)fminf(fmaxf(v, 0.f), 3.f);                      // boxes are valid (x2 >= x1 - 1): w*h >= 0
}

// One workgroup (1024 threads = 16 waves) per image; N <= 16384.  Output row order of an image:
//   level 0 rois (input order) | level 1 | level 2 | level 3 | input rows past n_valid (padding) | unused rows up to n_out,
// where with pad_empty a level without rois contributes ONE all-zero roi (perm -1), exactly the rows the reference's loader
// builds (core/rcnn.py:53-74).  Rows >= n_rows[b] are padding: zero boxes, level 0, never real rois.
__global__ __launch_bounds__(1024) void fpn_roi_dispatch_kernel(DispatchArgs g) {
  __shared__ int wave_cnt[16][16][5];        // [chunk][wave][level], level 4 = input rows past n_valid
  __shared__ int base[16][16][5];
  __shared__ int lvl_start[6];               // first output row of each group, [5] = first unused row
  __shared__ int lvl_cnt[5];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* rois = g.rois + (long)b * g.N * g.box_stride;
  const int chunks = (g.N + 1023) / 1024;
  const int nv = g.n_valid ? min(max(g.n_valid[b], 0), g.N) : g.N;
#pragma unroll 1
  for (int c = 0; c < chunks; ++c) {
    const int i = c * 1024 + tid;
    int lv = -1;
    if (i < g.N) {
      const float* r = rois + (long)i * g.box_stride + g.box_off;
      lv = i < nv ? fpn_level(r[0], r[1], r[2], r[3]) : 4;
    }
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const unsigned long long m = __ballot(lv == l);
      if (lane == 0) wave_cnt[c][wave][l] = __popcll(m);
    }
  }
  __syncthreads();
  if (tid < 5) {                              // thread l: size of group l
    int n = 0;
    for (int c = 0; c < chunks; ++c)
      for (int w = 0; w < 16; ++w) n += wave_cnt[c][w][tid];
    lvl_cnt[tid] = n;
    if (tid < 4) g.counts[b * 4 + tid] = n;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int l = 0; l < 5; ++l) {
      lvl_start[l] = run;
      run += lvl_cnt[l] + ((l < 4 && g.pad_empty && lvl_cnt[l] == 0) ? 1 : 0);
      if (l == 3 && g.n_rows) g.n_rows[b] = run;
    }
    lvl_start[5] = run;
  }
  __syncthreads();
  if (tid < 5) {                              // thread l: exclusive scan of group l over (chunk, wave)
    int run = lvl_start[tid];
    for (int c = 0; c < chunks; ++c)
      for (int w = 0; w < 16; ++w) { base[c][w][tid] = run; run += wave_cnt[c][w][tid]; }
  }
  __syncthreads();
  float* ro = g.rois_out + (long)b * g.n_out * 5;
  int* lo = g.level_out + (long)b * g.n_out;
  int* po = g.perm + (long)b * g.n_out;
  const float bidx = (float)(b + g.batch_index_base);
#pragma unroll 1
  for (int c = 0; c < chunks; ++c) {
    const int i = c * 1024 + tid;
    int lv = -1;
    const float* r = rois + (long)(i < g.N ? i : 0) * g.box_stride + g.box_off;
    float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
    if (i < g.N) lv = i < nv ? fpn_level(x1, y1, x2, y2) : 4;
    int pos = -1;
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const unsigned long long m = __ballot(lv == l);
      if (lv == l) pos = base[c][wave][l] + __popcll(m & ((1ull << lane) - 1ull));
    }
    if (i < g.N) {
      if (lv == 4) { x1 = y1 = x2 = y2 = 0.f; }
      float* o = ro + (long)pos * 5;
      o[0] = bidx; o[1] = x1; o[2] = y1; o[3] = x2; o[4] = y2;
      lo[pos] = lv == 4 ? 0 : lv;
      po[pos] = i;
    }
  }
  if (tid < 4 && g.pad_empty && lvl_cnt[tid] == 0) {          // the dummy roi of an empty level
    const int pos = lvl_start[tid];
    float* o = ro + (long)pos * 5;
    o[0] = bidx; o[1] = o[2] = o[3] = o[4] = 0.f;
    lo[pos] = tid; po[pos] = -1;
  }
  for (int pos = lvl_start[5] + tid; pos < g.n_out; pos += 1024) {   // rows nobody owns
    float* o = ro + (long)pos * 5;
    o[0] = bidx; o[1] = o[2] = o[3] = o[4] = 0.f;
    lo[pos] = 0; po[pos] = -1;
  }
}
#pragma clang fp contract(fast)

struct UpAddArgs {
  const void* top; void* lat;       // top [B, H/2, W/2, C], lat [B, H, W, C] (NHWC contiguous), in place on lat
  int B, H, W, C;
};

// thread = 8 channels (bf16) / 4 channels (fp32) of one lateral pixel
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_add_kernel(UpAddArgs g) {
  constexpr int V = 16 / sizeof(T);
  const int cv = g.C / V;
  const long total = (long)g.B * g.H * g.W * cv;
  const int Ht = g.H >> 1, Wt = g.W >> 1;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % cv);
    long r = idx / cv;
    const int x = (int)(r % g.W); r /= g.W;
    const int y = (int)(r % g.H);
    const int b = (int)(r / g.H);
    const uint4 a = *((const uint4*)g.lat + idx);
    const uint4 t = *((const uint4*)g.top + (((long)b * Ht + (y >> 1)) * Wt + (x >> 1)) * cv + c);
    uint4 o;
    if (sizeof(T) == 4) {
      const float* fa = (const float*)&a; const float* ft = (const float*)&t; float* fo = (float*)&o;
#pragma unroll
      for (int k = 0; k < 4; ++k) fo[k] = ft[k] + fa[k];
    } else {
      const unsigned int* ua = (const unsigned int*)&a; const unsigned int* ut = (const unsigned int*)&t;
      unsigned int* uo = (unsigned int*)&o;
#pragma unroll
      for (int k = 0; k < 4; ++k)
        uo[k] = pack_bf16x2(__uint_as_float(ut[k] << 16) + __uint_as_float(ua[k] << 16),
                            __uint_as_float(ut[k] & 0xffff0000u) + __uint_as_float(ua[k] & 0xffff0000u));
    }
    *((uint4*)g.lat + idx) = o;
  }
}

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_fpn_roi_dispatch_ex(const float* rois, int box_stride, int box_off, float* rois_out,
                                          int* level_out, int* perm, int* counts, int B, int N, int batch_index_base,
                                          const int* n_valid, int pad_empty, int n_out, int* n_rows, void* stream) {
  RELNET_REQUIRE(rois && rois_out && level_out && perm && counts, "relnet_fpn_roi_dispatch: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && N <= 16384, "relnet_fpn_roi_dispatch: need 0 < N <= 16384 rois per image, got %d", N);
  RELNET_REQUIRE(box_off >= 0 && box_off + 4 <= box_stride, "relnet_fpn_roi_dispatch: box_off %d / stride %d", box_off, box_stride);
  RELNET_REQUIRE(n_out >= N + (pad_empty ? 4 : 0), "relnet_fpn_roi_dispatch: n_out %d < N %d%s", n_out, N, pad_empty ? " + 4 dummy rows" : "");
  DispatchArgs g{rois, box_stride, box_off, rois_out, level_out, perm, counts, N, batch_index_base, n_valid, pad_empty, n_out, n_rows};
  fpn_roi_dispatch_kernel<<<B, 1024, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_fpn_roi_dispatch");
}

extern "C" int relnet_fpn_roi_dispatch(const float* rois, int box_stride, int box_off, float* rois_out,
                                       int* level_out, int* perm, int* counts, int B, int N,
                                       int batch_index_base, void* stream) {
  return relnet_fpn_roi_dispatch_ex(rois, box_stride, box_off, rois_out, level_out, perm, counts, B, N, batch_index_base,
                                    nullptr, 0, N, nullptr, stream);
}

extern "C" int relnet_upsample2x_add(const void* top, void* lateral, int B, int H, int W, int C, int dtype,
                                     void* stream) {
  RELNET_REQUIRE(top && lateral, "relnet_upsample2x_add: null operand");
  RELNET_REQUIRE(B > 0 && H > 0 && W > 0 && (H & 1) == 0 && (W & 1) == 0,
                 "relnet_upsample2x_add: lateral map %dx%d must be exactly twice the top map (IMAGE_STRIDE padding)", H, W);
  RELNET_REQUIRE(dtype == RELNET_F32 || dtype == RELNET_BF16, "relnet_upsample2x_add: unknown dtype %d", dtype);
  const int v = dtype == RELNET_BF16 ? 8 : 4;
  RELNET_REQUIRE(C % v == 0, "relnet_upsample2x_add: C %d must be a multiple of %d", C, v);
  UpAddArgs g{top, lateral, B, H, W, C};
  const long total = (long)B * H * W * (C / v);
  long blocks = (total + 255) / 256;
  if (blocks > 256L * 32) blocks = 256L * 32;
  if (dtype == RELNET_BF16) upsample2x_add_kernel<unsigned short><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(g);
  else upsample2x_add_kernel<float><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_upsample2x_add");
}
