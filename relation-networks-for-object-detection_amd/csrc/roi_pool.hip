// Max ROIPooling forward/backward (reference: mx.symbol.ROIPooling(pooled_size=(7,7),
// spatial_scale=1/16), SYM_REL:252-253; semantics of MXNet v1.1.0 src/operator/roi_pooling.cu).
// Element strides are explicit so the same kernel serves NCHW fp32 (the reference layout,
// parity tests) and channels-last bf16 (the throughput path, where the 256 channels of a
// bin are contiguous and a wavefront reads 64 consecutive channels per pixel).
#include "common.h"
#include <float.h>

namespace relnet {

struct RoiArgs {
  const void* data; long ds_b, ds_c, ds_h, ds_w;     // element strides of [B, C, H, W]
  const float* rois;                                  // [R, 5] batch_idx, x1, y1, x2, y2
  void* out; long os_r, os_c, os_ph, os_pw;           // element strides of [R, C, PH, PW]
  int* argmax;                                        // same strides as out, or nullptr
  int R, C, H, W, PH, PW, batch_index_base;
  float scale;
  // FPN (relnet_roi_pool_fpn_fwd): roi r pools from pyramid level level[r]; nullptr = single map above
  const int* level;
  struct Level { const void* data; long ds_b, ds_c, ds_h, ds_w; int H, W; float scale; } lv[4];
};

// Feature map of roi r as plain scalars (rois are level-sorted and r is uniform per workgroup, so this is a uniform
// kernarg read; the kernel-argument struct itself is never written -- doing so spills it to scratch, 6x slower).
struct RoiMap { const void* data; long ds_b, ds_c, ds_h, ds_w; int H, W; float scale; };
#define RELNET_SEL4(l, f) ((l) == 0 ? g.lv[0].f : (l) == 1 ? g.lv[1].f : (l) == 2 ? g.lv[2].f : g.lv[3].f)
__device__ __forceinline__ RoiMap roi_map(const RoiArgs& g, int r) {
  RoiMap m;
  if (g.level) {         // constant indices only: a dynamic index into the kernarg array forces a scratch copy of the struct
    const int l = g.level[r] & 3;
    m.data = RELNET_SEL4(l, data); m.ds_b = RELNET_SEL4(l, ds_b); m.ds_c = RELNET_SEL4(l, ds_c);
    m.ds_h = RELNET_SEL4(l, ds_h); m.ds_w = RELNET_SEL4(l, ds_w); m.H = RELNET_SEL4(l, H); m.W = RELNET_SEL4(l, W);
    m.scale = RELNET_SEL4(l, scale);
  } else {
    m.data = g.data; m.ds_b = g.ds_b; m.ds_c = g.ds_c; m.ds_h = g.ds_h; m.ds_w = g.ds_w; m.H = g.H; m.W = g.W; m.scale = g.scale;
  }
  return m;
}

template <typename T> __device__ __forceinline__ float ld(const T* p);
template <> __device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<unsigned short>(const unsigned short* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void st(T* p, float v);
template <> __device__ __forceinline__ void st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void st<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }

#pragma clang fp contract(off)
// grid.x = R * PH * PW bins, threads stride over channels
template <typename T>
__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(RoiArgs g) {
  const int bin = blockIdx.x;
  const int pw = bin % g.PW, ph = (bin / g.PW) % g.PH, r = bin / (g.PW * g.PH);
  const RoiMap m = roi_map(g, r);
  const float* roi = g.rois + (long)r * 5;
  const int b = (int)roi[0] - g.batch_index_base;
  const int rs_w = (int)roundf(roi[1] * m.scale), rs_h = (int)roundf(roi[2] * m.scale);
  const int re_w = (int)roundf(roi[3] * m.scale), re_h = (int)roundf(roi[4] * m.scale);
  const int rw = max(re_w - rs_w + 1, 1), rh = max(re_h - rs_h + 1, 1);   // malformed -> 1x1
  const float bin_h = (float)rh / (float)g.PH, bin_w = (float)rw / (float)g.PW;
  int hs = (int)floorf((float)ph * bin_h), he = (int)ceilf((float)(ph + 1) * bin_h);
  int ws = (int)floorf((float)pw * bin_w), we = (int)ceilf((float)(pw + 1) * bin_w);
  hs = min(max(hs + rs_h, 0), m.H); he = min(max(he + rs_h, 0), m.H);
  ws = min(max(ws + rs_w, 0), m.W); we = min(max(we + rs_w, 0), m.W);
  const bool empty = (he <= hs) || (we <= ws);
  const T* base = (const T*)m.data + (long)b * m.ds_b;
  for (int c = threadIdx.x; c < g.C; c += 256) {
    float best = empty ? 0.f : -FLT_MAX;
    int bi = -1;
    const T* pc = base + (long)c * m.ds_c;
    for (int y = hs; y < he; ++y)
      for (int x = ws; x < we; ++x) {
        const float v = ld<T>(pc + (long)y * m.ds_h + (long)x * m.ds_w);
        if (v > best) { best = v; bi = y * m.W + x; }
      }
    const long o = (long)r * g.os_r + (long)c * g.os_c + (long)ph * g.os_ph + (long)pw * g.os_pw;
    st<T>((T*)g.out + o, best);
    if (g.argmax) g.argmax[o] = bi;
  }
}
// Channels-last bf16 fast path: thread = (bin, 8-channel group); 16-byte loads, the 8 running
// maxima stay in registers.  Requires C % 8 == 0, ds_c == 1, os_c == 1 (bf16).
template <bool ARGMAX>
__global__ __launch_bounds__(256) void roi_pool_fwd_cl_kernel(RoiArgs g) {
  const int groups = g.C >> 3;                           // 8-channel groups per bin
  const int bins_per_blk = 256 / groups;                 // groups divides 256 (checked on the host)
  const int r = blockIdx.y;
  const int bin = blockIdx.x * bins_per_blk + threadIdx.x / groups;
  const int cg = threadIdx.x % groups;
  if (bin >= g.PH * g.PW) return;
  const int pw = bin % g.PW, ph = bin / g.PW;
  const RoiMap m = roi_map(g, r);
  const float* roi = g.rois + (long)r * 5;
  const int b = (int)roi[0] - g.batch_index_base;
  const int rs_w = (int)roundf(roi[1] * m.scale), rs_h = (int)roundf(roi[2] * m.scale);
  const int re_w = (int)roundf(roi[3] * m.scale), re_h = (int)roundf(roi[4] * m.scale);
  const int rw = max(re_w - rs_w + 1, 1), rh = max(re_h - rs_h + 1, 1);
  const float bin_h = (float)rh / (float)g.PH, bin_w = (float)rw / (float)g.PW;
  int hs = (int)floorf((float)ph * bin_h), he = (int)ceilf((float)(ph + 1) * bin_h);
  int ws = (int)floorf((float)pw * bin_w), we = (int)ceilf((float)(pw + 1) * bin_w);
  hs = min(max(hs + rs_h, 0), m.H); he = min(max(he + rs_h, 0), m.H);
  ws = min(max(ws + rs_w, 0), m.W); we = min(max(we + rs_w, 0), m.W);
  const bool empty = (he <= hs) || (we <= ws);
  float best[8];
  int bi[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { best[e] = empty ? 0.f : -FLT_MAX; bi[e] = -1; }
  const unsigned short* base = (const unsigned short*)m.data + (long)b * m.ds_b + cg * 8;
  for (int y = hs; y < he; ++y)
    for (int x = ws; x < we; ++x) {          // (measured: a 4-wide unrolled row walk is SLOWER, 449 vs 309 us at 54 images --
      const uint4 v = *(const uint4*)(base + (long)y * m.ds_h + (long)x * m.ds_w);      // most bins are 1-2 pixels wide)
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float lo = bf2f(w4[e] & 0xffff), hi = bf2f(w4[e] >> 16);
        if (lo > best[2 * e]) { best[2 * e] = lo; if (ARGMAX) bi[2 * e] = y * m.W + x; }
        if (hi > best[2 * e + 1]) { best[2 * e + 1] = hi; if (ARGMAX) bi[2 * e + 1] = y * m.W + x; }
      }
    }
  const long o = (long)r * g.os_r + (long)ph * g.os_ph + (long)pw * g.os_pw + cg * 8;
  *(uint4*)((unsigned short*)g.out + o) = make_uint4(pack_bf16x2(best[0], best[1]), pack_bf16x2(best[2], best[3]),
                                                     pack_bf16x2(best[4], best[5]), pack_bf16x2(best[6], best[7]));
  if (ARGMAX) {
#pragma unroll
    for (int e = 0; e < 8; ++e) g.argmax[o + e] = bi[e];
  }
}
#pragma clang fp contract(fast)

struct RoiBwdArgs {
  const void* grad_out; const int* argmax; long os_r, os_c, os_ph, os_pw;
  const float* rois;
  float* grad_in; long ds_b, ds_c;          // fp32 accumulation buffer (pre-zeroed): element (b, c, pixel a) at b ds_b + c ds_c + a ds_p
  int R, C, W, PH, PW, batch_index_base;
  const int* level;                         // FPN: roi r scatters into lv[level[r]]; nullptr = the single map above
  struct Level { float* grad_in; long ds_b, ds_c, ds_p; } lv[4];
  long ds_p;                                // pixel stride: 1 = [B,C,H*W] (NCHW); C with ds_c = 1 = [B,H*W,C] (NHWC: the lanes'
};                                          // consecutive channels hit consecutive words -> coalesced float atomics)

template <typename T>
__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(RoiBwdArgs g) {
  const int bin = blockIdx.x;
  const int pw = bin % g.PW, ph = (bin / g.PW) % g.PH, r = bin / (g.PW * g.PH);
  const int b = (int)g.rois[(long)r * 5] - g.batch_index_base;
  float* gin = g.grad_in; long ds_b = g.ds_b, ds_c = g.ds_c, ds_p = g.ds_p;
  if (g.level) {
    const int l = g.level[r] & 3;
    gin = RELNET_SEL4(l, grad_in); ds_b = RELNET_SEL4(l, ds_b); ds_c = RELNET_SEL4(l, ds_c); ds_p = RELNET_SEL4(l, ds_p);
  }
  for (int c = threadIdx.x; c < g.C; c += 256) {
    const long o = (long)r * g.os_r + (long)c * g.os_c + (long)ph * g.os_ph + (long)pw * g.os_pw;
    const int a = g.argmax[o];
    if (a >= 0) atomicAdd(gin + (long)b * ds_b + (long)c * ds_c + (long)a * ds_p, ld<T>((const T*)g.grad_out + o));
  }
}

}  // namespace relnet

using namespace relnet;
enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

static int launch_roi_pool(const RoiArgs& g, int dtype, bool aligned, void* stream, const char* what) {
  dim3 grid((unsigned)((long)g.R * g.PH * g.PW));
  const int groups = g.C / 8;
  if (dtype == RELNET_BF16 && g.C % 8 == 0 && groups <= 256 && 256 % groups == 0 && g.ds_c == 1 && g.os_c == 1 && aligned &&
      g.os_r % 8 == 0 && g.os_ph % 8 == 0 && g.os_pw % 8 == 0) {
    const int bins_per_blk = 256 / groups;
    dim3 g2((g.PH * g.PW + bins_per_blk - 1) / bins_per_blk, g.R);
    if (g.argmax) roi_pool_fwd_cl_kernel<true><<<g2, 256, 0, (hipStream_t)stream>>>(g);
    else roi_pool_fwd_cl_kernel<false><<<g2, 256, 0, (hipStream_t)stream>>>(g);
    return check_launch(what);
  }
  if (dtype == RELNET_F32) roi_pool_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else if (dtype == RELNET_BF16) roi_pool_fwd_kernel<unsigned short><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else RELNET_REQUIRE(false, "%s: unknown dtype %d", what, dtype);
  return check_launch(what);
}

extern "C" int relnet_roi_pool_fwd(const void* data, const long* data_strides4, const float* rois,
                                   void* out, const long* out_strides4, int* argmax, int R, int C,
                                   int H, int W, int PH, int PW, float spatial_scale,
                                   int batch_index_base, int dtype, void* stream) {
  RELNET_REQUIRE(data && rois && out && data_strides4 && out_strides4, "relnet_roi_pool_fwd: null operand");
  RELNET_REQUIRE(R > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "relnet_roi_pool_fwd: bad shape");
  RoiArgs g;
  g.data = data; g.ds_b = data_strides4[0]; g.ds_c = data_strides4[1]; g.ds_h = data_strides4[2]; g.ds_w = data_strides4[3];
  g.rois = rois; g.out = out; g.os_r = out_strides4[0]; g.os_c = out_strides4[1]; g.os_ph = out_strides4[2]; g.os_pw = out_strides4[3];
  g.argmax = argmax; g.R = R; g.C = C; g.H = H; g.W = W; g.PH = PH; g.PW = PW; g.scale = spatial_scale;
  g.batch_index_base = batch_index_base;
  g.level = nullptr;
  return launch_roi_pool(g, dtype, g.ds_h % 8 == 0 && g.ds_w % 8 == 0 && g.ds_b % 8 == 0, stream, "relnet_roi_pool_fwd");
}

extern "C" int relnet_roi_pool_fpn_fwd(const void* const* data_levels, const long* data_strides4_levels,
                                       const int* heights, const int* widths, const float* spatial_scales,
                                       int num_levels, const float* rois, const int* roi_level, void* out,
                                       const long* out_strides4, int* argmax, int R, int C, int PH, int PW,
                                       int batch_index_base, int dtype, void* stream) {
  RELNET_REQUIRE(data_levels && data_strides4_levels && heights && widths && spatial_scales && rois && roi_level && out &&
                 out_strides4, "relnet_roi_pool_fpn_fwd: null operand");
  RELNET_REQUIRE(num_levels >= 1 && num_levels <= 4, "relnet_roi_pool_fpn_fwd: 1..4 pyramid levels, got %d", num_levels);
  RELNET_REQUIRE(R > 0 && C > 0 && PH > 0 && PW > 0, "relnet_roi_pool_fpn_fwd: bad shape");
  RoiArgs g;
  bool aligned = true;
  for (int l = 0; l < 4; ++l) {
    const int s = l < num_levels ? l : num_levels - 1;
    RELNET_REQUIRE(data_levels[s] && heights[s] > 0 && widths[s] > 0, "relnet_roi_pool_fpn_fwd: bad level %d", s);
    g.lv[l].data = data_levels[s]; g.lv[l].ds_b = data_strides4_levels[4 * s]; g.lv[l].ds_c = data_strides4_levels[4 * s + 1];
    g.lv[l].ds_h = data_strides4_levels[4 * s + 2];
    g.lv[l].ds_w = data_strides4_levels[4 * s + 3]; g.lv[l].H = heights[s]; g.lv[l].W = widths[s]; g.lv[l].scale = spatial_scales[s];
    aligned = aligned && g.lv[l].ds_c == 1 && g.lv[l].ds_b % 8 == 0 && g.lv[l].ds_h % 8 == 0 && g.lv[l].ds_w % 8 == 0;
  }
  g.data = g.lv[0].data; g.ds_b = g.lv[0].ds_b; g.ds_c = data_strides4_levels[1]; g.ds_h = g.lv[0].ds_h; g.ds_w = g.lv[0].ds_w;
  g.H = g.lv[0].H; g.W = g.lv[0].W; g.scale = g.lv[0].scale;
  g.rois = rois; g.out = out; g.os_r = out_strides4[0]; g.os_c = out_strides4[1]; g.os_ph = out_strides4[2]; g.os_pw = out_strides4[3];
  g.argmax = argmax; g.R = R; g.C = C; g.PH = PH; g.PW = PW; g.batch_index_base = batch_index_base; g.level = roi_level;
  return launch_roi_pool(g, dtype, aligned, stream, "relnet_roi_pool_fpn_fwd");
}

// Owner form of the backward (round 6): the scatter above issues one memory-side float atomic per (roi, bin, channel) -- 31 M at 8 images x 308 rois, and the
// proposals of an image overlap, so most of them hit the same few hundred cells and serialise: 0.44 ms per training step.  Here ONE workgroup owns
// (image, CH consecutive channels): the whole H x W x CH gradient slab lives in LDS (38 x 63 x 8 floats = 77 KB), the workgroup walks every (roi, bin) of its
// image, adds with LDS atomics, and flushes the slab once with plain read-modify-writes -- no global atomics.  Channels-last operands (grad_out / argmax
// [R][PH][PW][C], gradient [B][H W][C]), dense maps of <= kOwnerMaxCells cells; everything else (NCHW, FPN levels, large maps) keeps the scatter kernel.
constexpr int kOwnerMaxCells = 4608;            // 4608 cells x 8 channels x 4 B = 147 KB of LDS

constexpr int kOwnerList = 1024;                // rois per pass of the compact list (4 KB of LDS next to the slab)

constexpr int kOwnerThreads = 1024;             // 16 wavefronts on the one workgroup a CU holds: the (roi, bin) walk is a chain of global loads -> LDS adds

template <typename T, int CH>
__global__ __launch_bounds__(kOwnerThreads) void roi_pool_bwd_owner_kernel(RoiBwdArgs g, int cells, int dbg) {
  constexpr int NT = kOwnerThreads;
  extern __shared__ float slab[];               // [CH][cells], then the roi list
  int* list = (int*)(slab + (long)cells * CH);
  __shared__ int cnt;
  const int chunks = g.C / CH;
  // Workgroup i runs on XCD i % 8 (observed dispatch rule).  A 128-byte line of argmax holds 4 channel groups of a (roi, bin), a line of the bf16
  // gradient 8: with group = i % chunks they sat on 4 / 8 different XCDs and every L2 fetched the line for 32 / 16 of its bytes (193 us per launch at
  // 8 images: ~1 GB through the fabric for 185 MB of operands).  Each XCD now owns a contiguous run of chunks / 8 groups of an image.
  int within = blockIdx.x % chunks;
  if (chunks % 8 == 0) within = (within & 7) * (chunks >> 3) + (within >> 3);
  const int b = blockIdx.x / chunks, c0 = within * CH;
  for (int i = threadIdx.x; i < cells * CH; i += NT) slab[i] = 0.f;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  const int bins = g.PH * g.PW;
  // argmax / gradient of one (listed roi, bin) pair for this workgroup's CH channels
  auto fetch = [&](int p, int (&a)[CH], float (&v)[CH]) {
    const int r = list[p / bins], bin = p % bins;
    const int ph = bin / g.PW, pw = bin - ph * g.PW;
    long o = (long)r * g.os_r + (long)ph * g.os_ph + (long)pw * g.os_pw + c0;           // (os_c == 1)
    if (dbg & 2) o = (long)(threadIdx.x & 63) * 8 + c0;                                   // (timing ablation: every load hits the same few lines)
    const int4 a0 = *(const int4*)(g.argmax + o);
    a[0] = a0.x; a[1] = a0.y; a[2] = a0.z; a[3] = a0.w;
    if constexpr (CH == 8) {
      const int4 a1 = *(const int4*)(g.argmax + o + 4);
      a[4] = a1.x; a[5] = a1.y; a[6] = a1.z; a[7] = a1.w;
    }
    if constexpr (CH == 8 && sizeof(T) == 2) {             // bf16: the eight gradients are one 16-byte load (host: os_* % 8 == 0 for this path)
      const uint4 w = *(const uint4*)((const unsigned short*)g.grad_out + o);
      const unsigned int w4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] = bf2f(w4[e] & 0xffff); v[2 * e + 1] = bf2f(w4[e] >> 16); }
    } else {
#pragma unroll
      for (int e = 0; e < CH; ++e) v[e] = ld<T>((const T*)g.grad_out + o + e);
    }
  };
  auto add = [&](const int (&a)[CH], const float (&v)[CH]) {
#pragma unroll
    for (int e = 0; e < CH; ++e)
      if (a[e] >= 0 && a[e] < cells && !(dbg & 1)) atomicAdd(&slab[e * cells + a[e]], v[e]);      // [CH][cells]: the lanes of one instruction (same e) spread over all banks
                                                                                   // ([cells][CH] put them on 8 of 64: 178 us per launch at 8 images)
  };
  // pass over the rois in segments of kOwnerList: the image's own rois go into a compact list (a scan over (roi, bin) PAIRS that skips 7 of 8 of them
  // one dependent load at a time was the first form: latency bound), then the (listed roi, bin) pairs are walked two at a time, loads first
  for (int base = 0; base < g.R; base += kOwnerList) {
    for (int r = base + threadIdx.x; r < min(base + kOwnerList, g.R); r += NT)
      if ((int)g.rois[(long)r * 5] - g.batch_index_base == b) list[atomicAdd(&cnt, 1)] = r;
    __syncthreads();
    const int n = cnt, np = n * bins;
    int p = threadIdx.x;
    for (; p + NT < np; p += 2 * NT) {
      int a0[CH], a1[CH];
      float v0[CH], v1[CH];
      fetch(p, a0, v0);
      fetch(p + NT, a1, v1);
      add(a0, v0);
      add(a1, v1);
    }
    if (p < np) {
      int a0[CH];
      float v0[CH];
      fetch(p, a0, v0);
      add(a0, v0);
    }
    __syncthreads();
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
  }
  float* gin = g.grad_in + (long)b * g.ds_b + c0;
  for (int i = threadIdx.x; i < cells; i += NT) {
    float* dst = gin + (long)i * g.ds_p;
#pragma unroll
    for (int q = 0; q < CH / 4; ++q) {
      float4 cur = *(float4*)(dst + 4 * q);
      cur.x += slab[(4 * q) * cells + i]; cur.y += slab[(4 * q + 1) * cells + i]; cur.z += slab[(4 * q + 2) * cells + i]; cur.w += slab[(4 * q + 3) * cells + i];
      *(float4*)(dst + 4 * q) = cur;
    }
  }
}

static int g_roi_bwd_mode = 0;       // test / measurement knob: 0 auto, 1 = scatter kernel only; 4 + bits = timing ablations of the owner kernel (results WRONG): 1 no LDS adds, 2 no scattered loads
extern "C" void relnet_roi_pool_bwd_debug(int mode) { g_roi_bwd_mode = mode; }

// Channels-last entry: grad_in fp32 [B][H][W][C] (dense, accumulated into), grad_out / argmax logical [R,C,PH,PW] with element strides out_strides4.
// Takes the owner form when it applies (channels contiguous in grad_out / argmax, C % 8 == 0, H W <= kOwnerMaxCells, 16-byte aligned rows), else the scatter kernel.
extern "C" int relnet_roi_pool_bwd_cl(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois, float* grad_in,
                                      int B, int H, int W, int R, int C, int PH, int PW, int batch_index_base, int dtype, void* stream) {
  RELNET_REQUIRE(grad_out && argmax && out_strides4 && rois && grad_in, "relnet_roi_pool_bwd_cl: null operand");
  RELNET_REQUIRE(B > 0 && H > 0 && W > 0 && R > 0 && C > 0 && PH > 0 && PW > 0, "relnet_roi_pool_bwd_cl: bad shape");
  RELNET_REQUIRE(dtype == RELNET_F32 || dtype == RELNET_BF16, "relnet_roi_pool_bwd_cl: unknown dtype %d", dtype);
  const long cells = (long)H * W;
  RoiBwdArgs g{grad_out, argmax, out_strides4[0], out_strides4[1], out_strides4[2], out_strides4[3],
               rois, grad_in, cells * C, 1, R, C, W, PH, PW, batch_index_base, nullptr, {}, (long)C};
  hipStream_t s = (hipStream_t)stream;
  // (one workgroup per CU at least: B C / 8 >= 256.  Same box, r06: 8 images 18.03 -> 17.81 ms with the owner form, ONE image 6.91 -> 7.00 ms -- 32 - 64 workgroups
  //  walking 308 rois lose to the 65 us scatter -- so small steps keep the scatter kernel)
  const bool owner = g_roi_bwd_mode != 1 && (long)B * C / 8 >= 256 && g.os_c == 1 && C % 8 == 0 && cells <= kOwnerMaxCells && g.os_r % 4 == 0 && g.os_ph % 4 == 0 &&
                     g.os_pw % 4 == 0 && (((uintptr_t)argmax) & 15) == 0 && (((uintptr_t)grad_in) & 15) == 0 &&
                     (dtype != RELNET_BF16 || (g.os_r % 8 == 0 && g.os_ph % 8 == 0 && g.os_pw % 8 == 0 && (((uintptr_t)grad_out) & 15) == 0));
  if (owner) {
    const unsigned grid = (unsigned)(B * (C / 8));
    const size_t lds = (size_t)cells * 8 * sizeof(float) + kOwnerList * sizeof(int);
    static relnet::PerDeviceOnce attr_once;
    if (attr_once.first()) {
      hipFuncSetAttribute((const void*)roi_pool_bwd_owner_kernel<float, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);      // (+ 4 B of static LDS: the list counter)
      hipFuncSetAttribute((const void*)roi_pool_bwd_owner_kernel<unsigned short, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    }
    if (dtype == RELNET_F32) roi_pool_bwd_owner_kernel<float, 8><<<grid, kOwnerThreads, lds, s>>>(g, (int)cells, g_roi_bwd_mode >= 4 ? g_roi_bwd_mode - 4 : 0);
    else roi_pool_bwd_owner_kernel<unsigned short, 8><<<grid, kOwnerThreads, lds, s>>>(g, (int)cells, g_roi_bwd_mode >= 4 ? g_roi_bwd_mode - 4 : 0);
    return check_launch("relnet_roi_pool_bwd_cl");
  }
  dim3 grid((unsigned)((long)R * PH * PW));
  if (dtype == RELNET_F32) roi_pool_bwd_kernel<float><<<grid, 256, 0, s>>>(g);
  else roi_pool_bwd_kernel<unsigned short><<<grid, 256, 0, s>>>(g);
  return check_launch("relnet_roi_pool_bwd_cl");
}

extern "C" int relnet_roi_pool_bwd_ex(const void* grad_out, const int* argmax, const long* out_strides4,
                                      const float* rois, float* grad_in, long gs_b, long gs_c, long gs_p, int R,
                                      int C, int W, int PH, int PW, int batch_index_base, int dtype,
                                      void* stream) {
  RELNET_REQUIRE(grad_out && argmax && rois && grad_in, "relnet_roi_pool_bwd: null operand");
  RoiBwdArgs g{grad_out, argmax, out_strides4[0], out_strides4[1], out_strides4[2], out_strides4[3],
               rois, grad_in, gs_b, gs_c, R, C, W, PH, PW, batch_index_base, nullptr, {}, gs_p};
  dim3 grid((unsigned)((long)R * PH * PW));
  if (dtype == RELNET_F32) roi_pool_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else if (dtype == RELNET_BF16) roi_pool_bwd_kernel<unsigned short><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else RELNET_REQUIRE(false, "relnet_roi_pool_bwd: unknown dtype %d", dtype);
  return check_launch("relnet_roi_pool_bwd");
}

extern "C" int relnet_roi_pool_bwd(const void* grad_out, const int* argmax, const long* out_strides4,
                                   const float* rois, float* grad_in, long gs_b, long gs_c, int R,
                                   int C, int W, int PH, int PW, int batch_index_base, int dtype,
                                   void* stream) {
  return relnet_roi_pool_bwd_ex(grad_out, argmax, out_strides4, rois, grad_in, gs_b, gs_c, 1, R, C, W, PH, PW,
                                batch_index_base, dtype, stream);
}

extern "C" int relnet_roi_pool_fpn_bwd_ex(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois,
                                          const int* roi_level, float* const* grad_in_levels, const long* gs_b_levels,
                                          const long* gs_c_levels, const long* gs_p_levels, int num_levels, int R, int C,
                                          int PH, int PW, int batch_index_base, int dtype, void* stream) {
  RELNET_REQUIRE(grad_out && argmax && out_strides4 && rois && roi_level && grad_in_levels && gs_b_levels && gs_c_levels,
                 "relnet_roi_pool_fpn_bwd: null operand");
  RELNET_REQUIRE(num_levels >= 1 && num_levels <= 4, "relnet_roi_pool_fpn_bwd: 1..4 pyramid levels, got %d", num_levels);
  RoiBwdArgs g{grad_out, argmax, out_strides4[0], out_strides4[1], out_strides4[2], out_strides4[3],
               rois, grad_in_levels[0], gs_b_levels[0], gs_c_levels[0], R, C, 0, PH, PW, batch_index_base, roi_level, {},
               gs_p_levels ? gs_p_levels[0] : 1};
  for (int l = 0; l < 4; ++l) {
    const int s = l < num_levels ? l : num_levels - 1;
    RELNET_REQUIRE(grad_in_levels[s], "relnet_roi_pool_fpn_bwd: null level %d", s);
    g.lv[l].grad_in = grad_in_levels[s]; g.lv[l].ds_b = gs_b_levels[s]; g.lv[l].ds_c = gs_c_levels[s];
    g.lv[l].ds_p = gs_p_levels ? gs_p_levels[s] : 1;
  }
  dim3 grid((unsigned)((long)R * PH * PW));
  if (dtype == RELNET_F32) roi_pool_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else if (dtype == RELNET_BF16) roi_pool_bwd_kernel<unsigned short><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else RELNET_REQUIRE(false, "relnet_roi_pool_fpn_bwd: unknown dtype %d", dtype);
  return check_launch("relnet_roi_pool_fpn_bwd");
}

// (An FPN form of the owner kernel -- per level, the stride-4 map cut into 12 bands of whole rows -- was built and measured in round 6: 37.2 ms per 8-image
//  FPN step with either kernel; the band workgroups re-read the level's argmax rows once per band, which costs what the contention did.  Removed.)
extern "C" int relnet_roi_pool_fpn_bwd(const void* grad_out, const int* argmax, const long* out_strides4, const float* rois,
                                       const int* roi_level, float* const* grad_in_levels, const long* gs_b_levels,
                                       const long* gs_c_levels, int num_levels, int R, int C, int PH, int PW,
                                       int batch_index_base, int dtype, void* stream) {
  return relnet_roi_pool_fpn_bwd_ex(grad_out, argmax, out_strides4, rois, roi_level, grad_in_levels, gs_b_levels, gs_c_levels,
                                    nullptr, num_levels, R, C, PH, PW, batch_index_base, dtype, stream);
}
