// ROIAlign forward / backward (named by BASELINE.json's north_star next to the proposal op; the reference graphs themselves pool with
// mx.symbol.ROIPooling -- symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py:252-253 -- and ship no ROIAlign, so
// this operator follows the PUBLISHED algorithm: He et al., "Mask R-CNN" (2017) section 3 as implemented by Detectron's RoIAlign /
// mx.contrib.sym.ROIAlign of MXNet >= 1.3: no coordinate rounding, sampling_ratio x sampling_ratio bilinear samples per bin, averaged).
//
//   roi_start = x1 * scale - off, roi_end = x2 * scale - off            (off = 0.5 when `aligned`, else 0)
//   roi_w = roi_end_w - roi_start_w (not aligned: max(., 1)), bin_w = roi_w / PW, grid_w = sampling_ratio > 0 ? sampling_ratio : ceil(roi_w / PW)
//   sample (iy, ix) of bin (ph, pw): y = roi_start_h + ph bin_h + (iy + 0.5) bin_h / grid_h,  x likewise
//   bilinear(y, x): 0 outside (-1, H) x (-1, W); coordinates clamped to [0, H - 1] x [0, W - 1]; out = sum / max(grid_h grid_w, 1)
//
// Explicit element strides like csrc/roi_pool.hip: NCHW fp32 (parity tests) and channels-last bf16 (thread = (bin, 8 channels), four 16-byte
// corner loads per sample).  Arithmetic in fp32 with contraction off, in the order oracle/roi_align.py restates: bit-identical on fp32 data.
#include "common.h"

namespace relnet {

struct RoiAlignArgs {
  const void* data; long ds_b, ds_c, ds_h, ds_w;     // element strides of [B, C, H, W]
  const float* rois;                                  // [R, 5] batch_idx, x1, y1, x2, y2
  void* out; long os_r, os_c, os_ph, os_pw;           // element strides of [R, C, PH, PW]  (backward: the output GRADIENT)
  float* grad_in; long gs_b, gs_c, gs_h, gs_w;        // backward: fp32 accumulation buffer (pre-zeroed), any layout
  int R, C, H, W, PH, PW, sampling_ratio, aligned, batch_index_base;
  float scale;
};

template <typename T> __device__ __forceinline__ float ra_ld(const T* p);
template <> __device__ __forceinline__ float ra_ld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ra_ld<unsigned short>(const unsigned short* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void ra_st(T* p, float v);
template <> __device__ __forceinline__ void ra_st<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void ra_st<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }

#pragma clang fp contract(off)
struct RoiBin { int b; float start_h, start_w, bin_h, bin_w; int grid_h, grid_w; float count; };

__device__ __forceinline__ RoiBin roi_bin(const RoiAlignArgs& g, int r) {
  const float* roi = g.rois + (long)r * 5;
  RoiBin q;
  q.b = (int)roi[0] - g.batch_index_base;
  const float off = g.aligned ? 0.5f : 0.f;
  q.start_w = roi[1] * g.scale - off; q.start_h = roi[2] * g.scale - off;
  const float end_w = roi[3] * g.scale - off, end_h = roi[4] * g.scale - off;
  float rw = end_w - q.start_w, rh = end_h - q.start_h;
  if (!g.aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
  q.bin_h = rh / (float)g.PH; q.bin_w = rw / (float)g.PW;
  q.grid_h = g.sampling_ratio > 0 ? g.sampling_ratio : (int)ceilf(rh / (float)g.PH);
  q.grid_w = g.sampling_ratio > 0 ? g.sampling_ratio : (int)ceilf(rw / (float)g.PW);
  q.count = fmaxf((float)(q.grid_h * q.grid_w), 1.f);
  return q;
}

// the four corners and weights of one sample; returns false when the sample lies outside the map (contributes 0)
struct Corners { int yl, xl, yh, xh; float w1, w2, w3, w4; };
__device__ __forceinline__ bool corners(float y, float x, int H, int W, Corners& c) {
  if (y < -1.f || y > (float)H || x < -1.f || x > (float)W) return false;
  if (y <= 0.f) y = 0.f;
  if (x <= 0.f) x = 0.f;
  c.yl = (int)y; c.xl = (int)x;
  if (c.yl >= H - 1) { c.yh = c.yl = H - 1; y = (float)c.yl; } else c.yh = c.yl + 1;
  if (c.xl >= W - 1) { c.xh = c.xl = W - 1; x = (float)c.xl; } else c.xh = c.xl + 1;
  const float ly = y - (float)c.yl, lx = x - (float)c.xl, hy = 1.f - ly, hx = 1.f - lx;
  c.w1 = hy * hx; c.w2 = hy * lx; c.w3 = ly * hx; c.w4 = ly * lx;
  return true;
}

// grid.x = R * PH * PW bins, threads stride over channels (any layout)
template <typename T>
__global__ __launch_bounds__(256) void roi_align_fwd_kernel(RoiAlignArgs g) {
  const int bin = blockIdx.x;
  const int pw = bin % g.PW, ph = (bin / g.PW) % g.PH, r = bin / (g.PW * g.PH);
  const RoiBin q = roi_bin(g, r);
  const T* base = (const T*)g.data + (long)q.b * g.ds_b;
  for (int c = threadIdx.x; c < g.C; c += 256) {
    const T* pc = base + (long)c * g.ds_c;
    float sum = 0.f;
    for (int iy = 0; iy < q.grid_h; ++iy) {
      const float y = q.start_h + (float)ph * q.bin_h + ((float)iy + 0.5f) * q.bin_h / (float)q.grid_h;
      for (int ix = 0; ix < q.grid_w; ++ix) {
        const float x = q.start_w + (float)pw * q.bin_w + ((float)ix + 0.5f) * q.bin_w / (float)q.grid_w;
        Corners k;
        if (!corners(y, x, g.H, g.W, k)) continue;
        const float v1 = ra_ld<T>(pc + (long)k.yl * g.ds_h + (long)k.xl * g.ds_w), v2 = ra_ld<T>(pc + (long)k.yl * g.ds_h + (long)k.xh * g.ds_w);
        const float v3 = ra_ld<T>(pc + (long)k.yh * g.ds_h + (long)k.xl * g.ds_w), v4 = ra_ld<T>(pc + (long)k.yh * g.ds_h + (long)k.xh * g.ds_w);
        sum += ((k.w1 * v1 + k.w2 * v2) + k.w3 * v3) + k.w4 * v4;
      }
    }
    ra_st<T>((T*)g.out + (long)r * g.os_r + (long)c * g.os_c + (long)ph * g.os_ph + (long)pw * g.os_pw, sum / q.count);
  }
}

// Channels-last bf16: thread = (bin, 8-channel group); a sample = four 16-byte corner loads.  C % 8 == 0, ds_c == os_c == 1.
__global__ __launch_bounds__(256) void roi_align_fwd_cl_kernel(RoiAlignArgs g) {
  const int groups = g.C >> 3;
  const int bins_per_blk = 256 / groups;
  const int r = blockIdx.y;
  const int bin = blockIdx.x * bins_per_blk + threadIdx.x / groups;
  const int cg = threadIdx.x % groups;
  if (bin >= g.PH * g.PW) return;
  const int pw = bin % g.PW, ph = bin / g.PW;
  const RoiBin q = roi_bin(g, r);
  const unsigned short* base = (const unsigned short*)g.data + (long)q.b * g.ds_b + cg * 8;
  float sum[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] = 0.f;
  for (int iy = 0; iy < q.grid_h; ++iy) {
    const float y = q.start_h + (float)ph * q.bin_h + ((float)iy + 0.5f) * q.bin_h / (float)q.grid_h;
    for (int ix = 0; ix < q.grid_w; ++ix) {
      const float x = q.start_w + (float)pw * q.bin_w + ((float)ix + 0.5f) * q.bin_w / (float)q.grid_w;
      Corners k;
      if (!corners(y, x, g.H, g.W, k)) continue;
      const uint4 a1 = *(const uint4*)(base + (long)k.yl * g.ds_h + (long)k.xl * g.ds_w), a2 = *(const uint4*)(base + (long)k.yl * g.ds_h + (long)k.xh * g.ds_w);
      const uint4 a3 = *(const uint4*)(base + (long)k.yh * g.ds_h + (long)k.xl * g.ds_w), a4 = *(const uint4*)(base + (long)k.yh * g.ds_h + (long)k.xh * g.ds_w);
      const unsigned int u1[4] = {a1.x, a1.y, a1.z, a1.w}, u2[4] = {a2.x, a2.y, a2.z, a2.w}, u3[4] = {a3.x, a3.y, a3.z, a3.w}, u4[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sum[2 * e] += ((k.w1 * bf2f(u1[e] & 0xffff) + k.w2 * bf2f(u2[e] & 0xffff)) + k.w3 * bf2f(u3[e] & 0xffff)) + k.w4 * bf2f(u4[e] & 0xffff);
        sum[2 * e + 1] += ((k.w1 * bf2f(u1[e] >> 16) + k.w2 * bf2f(u2[e] >> 16)) + k.w3 * bf2f(u3[e] >> 16)) + k.w4 * bf2f(u4[e] >> 16);
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sum[e] = sum[e] / q.count;
  const long o = (long)r * g.os_r + (long)ph * g.os_ph + (long)pw * g.os_pw + cg * 8;
  *(uint4*)((unsigned short*)g.out + o) = make_uint4(pack_bf16x2(sum[0], sum[1]), pack_bf16x2(sum[2], sum[3]),
                                                     pack_bf16x2(sum[4], sum[5]), pack_bf16x2(sum[6], sum[7]));
}

// Backward: every sample scatters w_k * dOut / count to its four corners (fp32 atomics; with a channels-last accumulation buffer the lanes'
// consecutive channels hit consecutive words).  grid.x = R * PH * PW bins, threads stride over channels.
template <typename T>
__global__ __launch_bounds__(256) void roi_align_bwd_kernel(RoiAlignArgs g) {
  const int bin = blockIdx.x;
  const int pw = bin % g.PW, ph = (bin / g.PW) % g.PH, r = bin / (g.PW * g.PH);
  const RoiBin q = roi_bin(g, r);
  float* gb = g.grad_in + (long)q.b * g.gs_b;
  for (int c = threadIdx.x; c < g.C; c += 256) {
    const float go = ra_ld<T>((const T*)g.out + (long)r * g.os_r + (long)c * g.os_c + (long)ph * g.os_ph + (long)pw * g.os_pw) / q.count;
    float* gc = gb + (long)c * g.gs_c;
    for (int iy = 0; iy < q.grid_h; ++iy) {
      const float y = q.start_h + (float)ph * q.bin_h + ((float)iy + 0.5f) * q.bin_h / (float)q.grid_h;
      for (int ix = 0; ix < q.grid_w; ++ix) {
        const float x = q.start_w + (float)pw * q.bin_w + ((float)ix + 0.5f) * q.bin_w / (float)q.grid_w;
        Corners k;
        if (!corners(y, x, g.H, g.W, k)) continue;
        atomicAdd(gc + (long)k.yl * g.gs_h + (long)k.xl * g.gs_w, k.w1 * go);
        atomicAdd(gc + (long)k.yl * g.gs_h + (long)k.xh * g.gs_w, k.w2 * go);
        atomicAdd(gc + (long)k.yh * g.gs_h + (long)k.xl * g.gs_w, k.w3 * go);
        atomicAdd(gc + (long)k.yh * g.gs_h + (long)k.xh * g.gs_w, k.w4 * go);
      }
    }
  }
}
#pragma clang fp contract(fast)

}  // namespace relnet

using namespace relnet;
enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

extern "C" int relnet_roi_align_fwd(const void* data, const long* data_strides4, const float* rois, void* out, const long* out_strides4,
                                    int R, int C, int H, int W, int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
                                    int batch_index_base, int dtype, void* stream) {
  RELNET_REQUIRE(data && rois && out && data_strides4 && out_strides4, "relnet_roi_align_fwd: null operand");
  RELNET_REQUIRE(R > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "relnet_roi_align_fwd: bad shape");
  RoiAlignArgs g{};
  g.data = data; g.ds_b = data_strides4[0]; g.ds_c = data_strides4[1]; g.ds_h = data_strides4[2]; g.ds_w = data_strides4[3];
  g.rois = rois; g.out = out; g.os_r = out_strides4[0]; g.os_c = out_strides4[1]; g.os_ph = out_strides4[2]; g.os_pw = out_strides4[3];
  g.R = R; g.C = C; g.H = H; g.W = W; g.PH = PH; g.PW = PW; g.sampling_ratio = sampling_ratio; g.aligned = aligned;
  g.batch_index_base = batch_index_base; g.scale = spatial_scale;
  const int groups = C / 8;
  if (dtype == RELNET_BF16 && C % 8 == 0 && groups <= 256 && 256 % groups == 0 && g.ds_c == 1 && g.os_c == 1 && g.ds_b % 8 == 0 && g.ds_h % 8 == 0 &&
      g.ds_w % 8 == 0 && g.os_r % 8 == 0 && g.os_ph % 8 == 0 && g.os_pw % 8 == 0 && (((uintptr_t)data | (uintptr_t)out) & 15) == 0) {
    const int bins_per_blk = 256 / groups;
    dim3 g2((PH * PW + bins_per_blk - 1) / bins_per_blk, R);
    roi_align_fwd_cl_kernel<<<g2, 256, 0, (hipStream_t)stream>>>(g);
    return check_launch("relnet_roi_align_fwd");
  }
  dim3 grid((unsigned)((long)R * PH * PW));
  if (dtype == RELNET_F32) roi_align_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else if (dtype == RELNET_BF16) roi_align_fwd_kernel<unsigned short><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else RELNET_REQUIRE(false, "relnet_roi_align_fwd: unknown dtype %d", dtype);
  return check_launch("relnet_roi_align_fwd");
}

extern "C" int relnet_roi_align_bwd(const void* grad_out, const long* out_strides4, const float* rois, float* grad_in,
                                    const long* grad_in_strides4, int R, int C, int H, int W, int PH, int PW, float spatial_scale,
                                    int sampling_ratio, int aligned, int batch_index_base, int dtype, void* stream) {
  RELNET_REQUIRE(grad_out && rois && grad_in && out_strides4 && grad_in_strides4, "relnet_roi_align_bwd: null operand");
  RELNET_REQUIRE(R > 0 && C > 0 && H > 0 && W > 0 && PH > 0 && PW > 0, "relnet_roi_align_bwd: bad shape");
  RoiAlignArgs g{};
  g.rois = rois; g.out = const_cast<void*>(grad_out); g.os_r = out_strides4[0]; g.os_c = out_strides4[1]; g.os_ph = out_strides4[2]; g.os_pw = out_strides4[3];
  g.grad_in = grad_in; g.gs_b = grad_in_strides4[0]; g.gs_c = grad_in_strides4[1]; g.gs_h = grad_in_strides4[2]; g.gs_w = grad_in_strides4[3];
  g.R = R; g.C = C; g.H = H; g.W = W; g.PH = PH; g.PW = PW; g.sampling_ratio = sampling_ratio; g.aligned = aligned;
  g.batch_index_base = batch_index_base; g.scale = spatial_scale;
  dim3 grid((unsigned)((long)R * PH * PW));
  if (dtype == RELNET_F32) roi_align_bwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else if (dtype == RELNET_BF16) roi_align_bwd_kernel<unsigned short><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else RELNET_REQUIRE(false, "relnet_roi_align_bwd: unknown dtype %d", dtype);
  return check_launch("relnet_roi_align_bwd");
}
