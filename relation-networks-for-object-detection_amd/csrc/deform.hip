// Deformable operators of the DCN configuration.
//
//  * relnet_deformable_im2col : the sampling half of DeformableConvolutionOp::Forward
//      (relation_rcnn/operator_cxx/deformable_convolution-inl.h:91-143, kernel
//       nn/deformable_im2col.cuh:215-262 with deformable_im2col_bilinear :76-113).
//      The GEMM half is relnet_gemm_nt on the column matrix, with BN / bias / ReLU fused there.
//  * relnet_deformable_psroi_pool_fwd : DeformablePSROIPoolForwardKernel
//      (relation_rcnn/operator_cxx/deformable_psroi_pooling.cu:51-138).
//
// Layout: element strides are explicit, so the same entry points serve the reference's NCHW fp32
// tensors (parity) and channels-last bf16 (throughput; 8 channels = 16 bytes per lane).  The column
// matrix is [B*Ho*Wo][KH*KW*C] with K ordered (tap, channel) -- the K order of the packed conv weights
// [Cout][KH][KW][Cin], so the GEMM needs no transposes.
//
// All coordinate arithmetic is separately rounded fp32 (contract off) in the order the reference writes
// it; oracle/deform.py is the matching CPU restatement.
#include "common.h"

namespace relnet {

enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

struct DeformColArgs {
  const void* data; long ds_b, ds_c, ds_h, ds_w;        // [B,C,H,W] element strides
  const float* offset; long fs_b, fs_c, fs_h, fs_w;     // [B, 2*KH*KW*DG, Ho, Wo]
  void* col; long col_ld;                               // [B*Ho*Wo][col_ld], K = KH*KW*C used
  int B, C, H, W, Ho, Wo, KH, KW, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, DG;
};

template <typename T> __device__ __forceinline__ float dld(const T* p);
template <> __device__ __forceinline__ float dld<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float dld<unsigned short>(const unsigned short* p) { return bf2f(*p); }
template <typename T> __device__ __forceinline__ void dst(T* p, float v);
template <> __device__ __forceinline__ void dst<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void dst<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }

#pragma clang fp contract(off)

// Bilinear taps of one (output pixel, kernel tap, deformable group): deformable_im2col.cuh:241-252 + :76-113.
struct Taps {
  bool inside;
  int ya, yb, xa, xb;        // absolute rows / columns of the 4 corners
  float w1, w2, w3, w4;
};
__device__ __forceinline__ Taps deform_taps(const DeformColArgs& g, int b, int ho, int wo, int i, int j, int dgi) {
  Taps t;
  const int tap = i * g.KW + j;
  const float* po = g.offset + (long)b * g.fs_b + (long)ho * g.fs_h + (long)wo * g.fs_w +
                    (long)(dgi * 2 * g.KH * g.KW + 2 * tap) * g.fs_c;
  const float off_h = po[0], off_w = po[g.fs_c];
  const int h_in = ho * g.stride_h - g.pad_h, w_in = wo * g.stride_w - g.pad_w;
  const float h_im = (float)(h_in + i * g.dil_h) + off_h;
  const float w_im = (float)(w_in + j * g.dil_w) + off_w;
  t.inside = (h_im >= 0.f) && (w_im >= 0.f) && (h_im < (float)g.H) && (w_im < (float)g.W);
  float mh = (float)(i * g.dil_h) + off_h, mw = (float)(j * g.dil_w) + off_w;
  const int cur_h = g.H - h_in, cur_w = g.W - w_in;
  int h_low = (int)floorf(mh), w_low = (int)floorf(mw), h_high, w_high;
  if (h_low >= cur_h - 1) { h_high = h_low = cur_h - 1; mh = (float)h_low; } else h_high = h_low + 1;
  if (w_low >= cur_w - 1) { w_high = w_low = cur_w - 1; mw = (float)w_low; } else w_high = w_low + 1;
  const float lh = mh - (float)h_low, lw = mw - (float)w_low;
  const float hh = 1.f - lh, hw = 1.f - lw;
  t.w1 = hh * hw; t.w2 = hh * lw; t.w3 = lh * hw; t.w4 = lh * lw;
  t.ya = min(max(h_in + h_low, 0), g.H - 1); t.yb = min(max(h_in + h_high, 0), g.H - 1);
  t.xa = min(max(w_in + w_low, 0), g.W - 1); t.xb = min(max(w_in + w_high, 0), g.W - 1);
  return t;
}

// Generic strides: one thread per (pixel, tap, channel), channel fastest.
template <typename TIN, typename TCOL>
__global__ __launch_bounds__(256) void deformable_im2col_kernel(DeformColArgs g) {
  const long total = (long)g.B * g.Ho * g.Wo * g.KH * g.KW * g.C;
  const int cpg = g.C / g.DG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int c = (int)(idx % g.C);
    long r = idx / g.C;
    const int tap = (int)(r % (g.KH * g.KW)); r /= (g.KH * g.KW);
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int ho = (int)(r % g.Ho);
    const int b = (int)(r / g.Ho);
    const Taps t = deform_taps(g, b, ho, wo, tap / g.KW, tap % g.KW, c / cpg);
    float val = 0.f;
    if (t.inside) {
      const TIN* p = (const TIN*)g.data + (long)b * g.ds_b + (long)c * g.ds_c;
      const float v1 = dld<TIN>(p + (long)t.ya * g.ds_h + (long)t.xa * g.ds_w);
      const float v2 = dld<TIN>(p + (long)t.ya * g.ds_h + (long)t.xb * g.ds_w);
      const float v3 = dld<TIN>(p + (long)t.yb * g.ds_h + (long)t.xa * g.ds_w);
      const float v4 = dld<TIN>(p + (long)t.yb * g.ds_h + (long)t.xb * g.ds_w);
      val = t.w1 * v1 + t.w2 * v2 + t.w3 * v3 + t.w4 * v4;
    }
    const long row = ((long)b * g.Ho + ho) * g.Wo + wo;
    dst<TCOL>((TCOL*)g.col + row * g.col_ld + (long)tap * g.C + c, val);
  }
}

// Channels-last bf16: one thread per (pixel, tap, 8-channel chunk); four 16-byte corner loads, one
// 16-byte store.  Requires ds_c == 1, C % 8 == 0, (C / DG) % 8 == 0 and 16-byte aligned strides.
__global__ __launch_bounds__(256) void deformable_im2col_cl_kernel(DeformColArgs g) {
  const int chunks = g.C / 8;
  const long total = (long)g.B * g.Ho * g.Wo * g.KH * g.KW * chunks;
  const int cpg = g.C / g.DG;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cc = (int)(idx % chunks);
    long r = idx / chunks;
    const int tap = (int)(r % (g.KH * g.KW)); r /= (g.KH * g.KW);
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int ho = (int)(r % g.Ho);
    const int b = (int)(r / g.Ho);
    const int c = cc * 8;
    const Taps t = deform_taps(g, b, ho, wo, tap / g.KW, tap % g.KW, c / cpg);
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (t.inside) {
      const unsigned short* p = (const unsigned short*)g.data + (long)b * g.ds_b + c;
      const uint4 q1 = *(const uint4*)(p + (long)t.ya * g.ds_h + (long)t.xa * g.ds_w);
      const uint4 q2 = *(const uint4*)(p + (long)t.ya * g.ds_h + (long)t.xb * g.ds_w);
      const uint4 q3 = *(const uint4*)(p + (long)t.yb * g.ds_h + (long)t.xa * g.ds_w);
      const uint4 q4 = *(const uint4*)(p + (long)t.yb * g.ds_h + (long)t.xb * g.ds_w);
      const unsigned int* a1 = (const unsigned int*)&q1; const unsigned int* a2 = (const unsigned int*)&q2;
      const unsigned int* a3 = (const unsigned int*)&q3; const unsigned int* a4 = (const unsigned int*)&q4;
      unsigned int* po = (unsigned int*)&o;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float lo = t.w1 * __uint_as_float(a1[k] << 16) + t.w2 * __uint_as_float(a2[k] << 16) +
                         t.w3 * __uint_as_float(a3[k] << 16) + t.w4 * __uint_as_float(a4[k] << 16);
        const float hi = t.w1 * __uint_as_float(a1[k] & 0xffff0000u) + t.w2 * __uint_as_float(a2[k] & 0xffff0000u) +
                         t.w3 * __uint_as_float(a3[k] & 0xffff0000u) + t.w4 * __uint_as_float(a4[k] & 0xffff0000u);
        po[k] = pack_bf16x2(lo, hi);
      }
    }
    const long row = ((long)b * g.Ho + ho) * g.Wo + wo;
    *(uint4*)((unsigned short*)g.col + row * g.col_ld + (long)tap * g.C + c) = o;
  }
}

// (measured and removed, r03 / r04: a kernel with the bilinear sampling as the A-tile producer of the MFMA GEMM -- no column matrix in
//  HBM -- took 0.54 ms per res5 layer at 27 images against 0.33 ms sampling + 0.23 ms GEMM, the four-corner gather through L2 being the
//  bound of both forms; with the hand-scheduled GEMM loop of round 4 the two-kernel path is the faster one)


// ---------------------------------------------------------------------------------------------
struct PsroiArgs {
  const void* data; long ds_b, ds_c, ds_h, ds_w;       // [B, output_dim*group^2, H, W]
  const float* rois;                                   // [R,5]
  const float* trans;                                  // [R, 2*num_classes, part, part] or nullptr (no_trans)
  void* out; long os_r, os_c, os_ph, os_pw;            // [R, output_dim, P, P]
  float* top_count;                                    // same strides as out, or nullptr
  int R, H, W, output_dim, group, P, part, spp, num_classes, ch_each, batch_index_base;
  float scale, trans_std;
};

struct RoiGeom { float start_w, start_h, roi_w, roi_h, bin_w, bin_h, sub_w, sub_h; int b; };
__device__ __forceinline__ RoiGeom psroi_geom(const PsroiArgs& g, int n) {
  const float* roi = g.rois + (long)n * 5;
  RoiGeom q;
  q.b = (int)roi[0] - g.batch_index_base;
  q.start_w = roundf(roi[1]) * g.scale - 0.5f;                  // :67-70
  q.start_h = roundf(roi[2]) * g.scale - 0.5f;
  const float end_w = (roundf(roi[3]) + 1.f) * g.scale - 0.5f;
  const float end_h = (roundf(roi[4]) + 1.f) * g.scale - 0.5f;
  q.roi_w = fmaxf(end_w - q.start_w, 0.1f);                     // :73-74
  q.roi_h = fmaxf(end_h - q.start_h, 0.1f);
  q.bin_h = q.roi_h / (float)g.P; q.bin_w = q.roi_w / (float)g.P;
  q.sub_h = q.bin_h / (float)g.spp; q.sub_w = q.bin_w / (float)g.spp;
  return q;
}

// One thread per (roi, ph, pw, ctop), ctop fastest.
template <typename T>
__global__ __launch_bounds__(256) void deformable_psroi_pool_fwd_kernel(PsroiArgs g) {
  const long total = (long)g.R * g.P * g.P * g.output_dim;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int ctop = (int)(idx % g.output_dim);
    long r = idx / g.output_dim;
    const int pw = (int)(r % g.P); r /= g.P;
    const int ph = (int)(r % g.P);
    const int n = (int)(r / g.P);
    const RoiGeom q = psroi_geom(g, n);
    const int part_h = (int)floorf((float)ph / (float)g.P * (float)g.part);       // :92-93
    const int part_w = (int)floorf((float)pw / (float)g.P * (float)g.part);
    float tx = 0.f, ty = 0.f;
    if (g.trans) {
      const int cls = ctop / g.ch_each;
      const float* pt = g.trans + ((((long)n * g.num_classes + cls) * 2) * g.part + part_h) * g.part + part_w;
      tx = pt[0] * g.trans_std;
      ty = pt[(long)g.part * g.part] * g.trans_std;
    }
    float wstart = (float)pw * q.bin_w + q.start_w; wstart = wstart + tx * q.roi_w;    // :100-105
    float hstart = (float)ph * q.bin_h + q.start_h; hstart = hstart + ty * q.roi_h;
    int gw = (int)floorf((float)pw * (float)g.group / (float)g.P);
    int gh = (int)floorf((float)ph * (float)g.group / (float)g.P);
    gw = min(max(gw, 0), g.group - 1); gh = min(max(gh, 0), g.group - 1);
    const int c = (ctop * g.group + gh) * g.group + gw;
    const T* pc = (const T*)g.data + (long)q.b * g.ds_b + (long)c * g.ds_c;
    float sum = 0.f; int count = 0;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        float w = wstart + (float)iw * q.sub_w;
        float h = hstart + (float)ih * q.sub_h;
        if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
        const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
        const float dx = w - (float)x1, dy = h - (float)y1;
        const float v11 = dld<T>(pc + (long)y1 * g.ds_h + (long)x1 * g.ds_w);
        const float v12 = dld<T>(pc + (long)y2 * g.ds_h + (long)x1 * g.ds_w);
        const float v21 = dld<T>(pc + (long)y1 * g.ds_h + (long)x2 * g.ds_w);
        const float v22 = dld<T>(pc + (long)y2 * g.ds_h + (long)x2 * g.ds_w);
        const float val = (1.f - dx) * (1.f - dy) * v11 + (1.f - dx) * dy * v12 + dx * (1.f - dy) * v21 + dx * dy * v22;
        sum = sum + val;
        ++count;
      }
    const long o = (long)n * g.os_r + (long)ctop * g.os_c + (long)ph * g.os_ph + (long)pw * g.os_pw;
    dst<T>((T*)g.out + o, count == 0 ? 0.f : sum / (float)count);
    if (g.top_count) g.top_count[o] = (float)count;
  }
}

// Channels-last bf16, group_size 1, one class of offsets: thread = (roi, bin, 8-channel chunk).
__global__ __launch_bounds__(256) void deformable_psroi_pool_fwd_cl_kernel(PsroiArgs g) {
  const int chunks = g.output_dim / 8;
  const long total = (long)g.R * g.P * g.P * chunks;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cc = (int)(idx % chunks);
    long r = idx / chunks;
    const int pw = (int)(r % g.P); r /= g.P;
    const int ph = (int)(r % g.P);
    const int n = (int)(r / g.P);
    const RoiGeom q = psroi_geom(g, n);
    const int part_h = (int)floorf((float)ph / (float)g.P * (float)g.part);
    const int part_w = (int)floorf((float)pw / (float)g.P * (float)g.part);
    float tx = 0.f, ty = 0.f;
    if (g.trans) {
      const float* pt = g.trans + (((long)n * 2) * g.part + part_h) * g.part + part_w;
      tx = pt[0] * g.trans_std;
      ty = pt[(long)g.part * g.part] * g.trans_std;
    }
    float wstart = (float)pw * q.bin_w + q.start_w; wstart = wstart + tx * q.roi_w;
    float hstart = (float)ph * q.bin_h + q.start_h; hstart = hstart + ty * q.roi_h;
    const unsigned short* pc = (const unsigned short*)g.data + (long)q.b * g.ds_b + cc * 8;
    float sum[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sum[k] = 0.f;
    int count = 0;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        float w = wstart + (float)iw * q.sub_w;
        float h = hstart + (float)ih * q.sub_h;
        if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
        const int x1 = (int)floorf(w), x2 = (int)ceilf(w), y1 = (int)floorf(h), y2 = (int)ceilf(h);
        const float dx = w - (float)x1, dy = h - (float)y1;
        const float c11 = (1.f - dx) * (1.f - dy), c12 = (1.f - dx) * dy, c21 = dx * (1.f - dy), c22 = dx * dy;
        const uint4 q11 = *(const uint4*)(pc + (long)y1 * g.ds_h + (long)x1 * g.ds_w);
        const uint4 q12 = *(const uint4*)(pc + (long)y2 * g.ds_h + (long)x1 * g.ds_w);
        const uint4 q21 = *(const uint4*)(pc + (long)y1 * g.ds_h + (long)x2 * g.ds_w);
        const uint4 q22 = *(const uint4*)(pc + (long)y2 * g.ds_h + (long)x2 * g.ds_w);
        const unsigned int* a = (const unsigned int*)&q11; const unsigned int* bq = (const unsigned int*)&q12;
        const unsigned int* cq = (const unsigned int*)&q21; const unsigned int* d = (const unsigned int*)&q22;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float lo = c11 * __uint_as_float(a[k] << 16) + c12 * __uint_as_float(bq[k] << 16) +
                           c21 * __uint_as_float(cq[k] << 16) + c22 * __uint_as_float(d[k] << 16);
          const float hi = c11 * __uint_as_float(a[k] & 0xffff0000u) + c12 * __uint_as_float(bq[k] & 0xffff0000u) +
                           c21 * __uint_as_float(cq[k] & 0xffff0000u) + c22 * __uint_as_float(d[k] & 0xffff0000u);
          sum[2 * k] = sum[2 * k] + lo;
          sum[2 * k + 1] = sum[2 * k + 1] + hi;
        }
        ++count;
      }
    uint4 o4;
    unsigned int* po = (unsigned int*)&o4;
    const float fc = (float)max(count, 1);
#pragma unroll
    for (int k = 0; k < 4; ++k)
      po[k] = pack_bf16x2(count == 0 ? 0.f : sum[2 * k] / fc, count == 0 ? 0.f : sum[2 * k + 1] / fc);
    const long o = (long)n * g.os_r + (long)cc * 8 + (long)ph * g.os_ph + (long)pw * g.os_pw;
    *(uint4*)((unsigned short*)g.out + o) = o4;
  }
}

static inline unsigned grid_for(long total) {
  long blocks = (total + 255) / 256;
  const long cap = 256L * 64;              // grid-stride beyond 64 blocks per CU
  return (unsigned)(blocks < cap ? blocks : cap);
}

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_deformable_im2col(const void* data, const long* data_strides4, const float* offset,
                                        const long* offset_strides4, void* col, long col_ld, int B, int C,
                                        int H, int W, int KH, int KW, int pad_h, int pad_w, int stride_h,
                                        int stride_w, int dil_h, int dil_w, int num_deformable_group,
                                        int data_dtype, int col_dtype, void* stream) {
  RELNET_REQUIRE(data && offset && col && data_strides4 && offset_strides4, "relnet_deformable_im2col: null operand");
  RELNET_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && stride_h > 0 && stride_w > 0 &&
                 dil_h > 0 && dil_w > 0 && pad_h >= 0 && pad_w >= 0, "relnet_deformable_im2col: bad shape");
  RELNET_REQUIRE(num_deformable_group > 0 && C % num_deformable_group == 0,
                 "relnet_deformable_im2col: input channels %d must divide deformable group size %d", C, num_deformable_group);
  DeformColArgs g;
  g.data = data; g.ds_b = data_strides4[0]; g.ds_c = data_strides4[1]; g.ds_h = data_strides4[2]; g.ds_w = data_strides4[3];
  g.offset = offset; g.fs_b = offset_strides4[0]; g.fs_c = offset_strides4[1]; g.fs_h = offset_strides4[2]; g.fs_w = offset_strides4[3];
  g.col = col; g.col_ld = col_ld;
  g.B = B; g.C = C; g.H = H; g.W = W; g.KH = KH; g.KW = KW; g.pad_h = pad_h; g.pad_w = pad_w;
  g.stride_h = stride_h; g.stride_w = stride_w; g.dil_h = dil_h; g.dil_w = dil_w; g.DG = num_deformable_group;
  g.Ho = (H + 2 * pad_h - (dil_h * (KH - 1) + 1)) / stride_h + 1;
  g.Wo = (W + 2 * pad_w - (dil_w * (KW - 1) + 1)) / stride_w + 1;
  RELNET_REQUIRE(g.Ho > 0 && g.Wo > 0, "relnet_deformable_im2col: empty output");
  RELNET_REQUIRE(col_ld >= (long)KH * KW * C, "relnet_deformable_im2col: col_ld %ld < K %ld", col_ld, (long)KH * KW * C);
  hipStream_t s = (hipStream_t)stream;
  const long pixels = (long)B * g.Ho * g.Wo * KH * KW;
  if (data_dtype == RELNET_BF16 && col_dtype == RELNET_BF16 && g.ds_c == 1 && C % 8 == 0 &&
      (C / num_deformable_group) % 8 == 0 && g.ds_h % 8 == 0 && g.ds_w % 8 == 0 && g.ds_b % 8 == 0 && col_ld % 8 == 0 &&
      ((uintptr_t)data & 15) == 0 && ((uintptr_t)col & 15) == 0) {
    deformable_im2col_cl_kernel<<<grid_for(pixels * (C / 8)), 256, 0, s>>>(g);
    return check_launch("relnet_deformable_im2col");
  }
  const unsigned grid = grid_for(pixels * C);
  if (data_dtype == RELNET_F32 && col_dtype == RELNET_F32) deformable_im2col_kernel<float, float><<<grid, 256, 0, s>>>(g);
  else if (data_dtype == RELNET_F32 && col_dtype == RELNET_BF16) deformable_im2col_kernel<float, unsigned short><<<grid, 256, 0, s>>>(g);
  else if (data_dtype == RELNET_BF16 && col_dtype == RELNET_BF16) deformable_im2col_kernel<unsigned short, unsigned short><<<grid, 256, 0, s>>>(g);
  else if (data_dtype == RELNET_BF16 && col_dtype == RELNET_F32) deformable_im2col_kernel<unsigned short, float><<<grid, 256, 0, s>>>(g);
  else RELNET_REQUIRE(false, "relnet_deformable_im2col: unknown dtype %d/%d", data_dtype, col_dtype);
  return check_launch("relnet_deformable_im2col");
}

extern "C" int relnet_deformable_psroi_pool_fwd(const void* data, const long* data_strides4, const float* rois,
                                                const float* trans, void* out, const long* out_strides4,
                                                float* top_count, int R, int C, int H, int W, int output_dim,
                                                int group_size, int pooled_size, int part_size,
                                                int sample_per_part, float spatial_scale, float trans_std,
                                                int num_classes, int batch_index_base, int dtype, void* stream) {
  RELNET_REQUIRE(data && rois && out && data_strides4 && out_strides4, "relnet_deformable_psroi_pool_fwd: null operand");
  RELNET_REQUIRE(R > 0 && C > 0 && H > 0 && W > 0 && output_dim > 0 && group_size > 0 && pooled_size > 0 &&
                 sample_per_part > 0, "relnet_deformable_psroi_pool_fwd: bad shape");
  RELNET_REQUIRE(C == output_dim * group_size * group_size,
                 "relnet_deformable_psroi_pool_fwd: data channels %d != output_dim*group_size^2 = %d", C,
                 output_dim * group_size * group_size);
  const bool no_trans = (trans == nullptr);
  RELNET_REQUIRE(no_trans || (num_classes > 0 && output_dim % num_classes == 0),
                 "relnet_deformable_psroi_pool_fwd: output_dim %d not divisible by num_classes %d", output_dim, num_classes);
  PsroiArgs g;
  g.data = data; g.ds_b = data_strides4[0]; g.ds_c = data_strides4[1]; g.ds_h = data_strides4[2]; g.ds_w = data_strides4[3];
  g.rois = rois; g.trans = trans; g.out = out;
  g.os_r = out_strides4[0]; g.os_c = out_strides4[1]; g.os_ph = out_strides4[2]; g.os_pw = out_strides4[3];
  g.top_count = top_count; g.R = R; g.H = H; g.W = W; g.output_dim = output_dim; g.group = group_size;
  g.P = pooled_size; g.part = part_size > 0 ? part_size : pooled_size; g.spp = sample_per_part;
  g.num_classes = no_trans ? 1 : num_classes; g.ch_each = no_trans ? output_dim : output_dim / num_classes;
  g.batch_index_base = batch_index_base; g.scale = spatial_scale; g.trans_std = trans_std;
  hipStream_t s = (hipStream_t)stream;
  const long bins = (long)R * pooled_size * pooled_size;
  if (dtype == RELNET_BF16 && !top_count && group_size == 1 && g.num_classes == 1 && output_dim % 8 == 0 && g.ds_c == 1 &&
      g.os_c == 1 && g.ds_h % 8 == 0 && g.ds_w % 8 == 0 && g.ds_b % 8 == 0 && g.os_r % 8 == 0 && g.os_ph % 8 == 0 &&
      g.os_pw % 8 == 0 && ((uintptr_t)data & 15) == 0 && ((uintptr_t)out & 15) == 0) {
    deformable_psroi_pool_fwd_cl_kernel<<<grid_for(bins * (output_dim / 8)), 256, 0, s>>>(g);
    return check_launch("relnet_deformable_psroi_pool_fwd");
  }
  const unsigned grid = grid_for(bins * output_dim);
  if (dtype == RELNET_F32) deformable_psroi_pool_fwd_kernel<float><<<grid, 256, 0, s>>>(g);
  else if (dtype == RELNET_BF16) deformable_psroi_pool_fwd_kernel<unsigned short><<<grid, 256, 0, s>>>(g);
  else RELNET_REQUIRE(false, "relnet_deformable_psroi_pool_fwd: unknown dtype %d", dtype);
  return check_launch("relnet_deformable_psroi_pool_fwd");
}

// =====================================================================================================
// Backward (training of the DCN configuration).  Reference: DeformableConvolutionOp::Backward
// (deformable_convolution-inl.h:145-237) = col-gradient GEMM + deformable_col2im (nn/deformable_im2col.cuh:313-351,
// weights get_gradient_weight :114-158) + deformable_col2im_coord (:420-470, get_coordinate_weight :161-213), and
// DeformablePSROIPoolBackwardAccKernel (deformable_psroi_pooling.cu:178-285).  Implemented as the exact adjoint of
// the forward kernels above (same taps, same border rules), which is what the reference's formulas evaluate to
// wherever the forward is differentiable.
// =====================================================================================================
namespace relnet {

struct DeformBwdArgs {
  DeformColArgs f;                 // forward geometry: data (values, for the offset gradient), offset, shapes
  const void* dcol; long dcol_ld;  // [B*Ho*Wo][dcol_ld] gradient of the column matrix, column (tap*C + c)
  float* grad_data; long gs_b, gs_c, gs_h, gs_w;      // fp32, += (atomics), logical [B,C,H,W]
  float* grad_offset; long os_b, os_c, os_h, os_w;    // fp32, += (atomics), logical [B, 2*KH*KW*DG, Ho, Wo]
  int dcol_f32;
  int win;                         // gather form: window radius D (a pair is NEAR when its four corners lie within D cells of its undeformed tap position)
};

__device__ __forceinline__ bool pair_near(const Taps& t, int cy, int cx, int D) {
  return abs(t.ya - cy) <= D && abs(t.yb - cy) <= D && abs(t.xa - cx) <= D && abs(t.xb - cx) <= D;
}

// one thread per (pixel, tap, channel), channel fastest: a wavefront covers 64 consecutive channels of one
// (pixel, tap); when they share a deformable group the two offset gradients are reduced in-wave first.
template <typename TIN>
__global__ __launch_bounds__(256) void deformable_col2im_kernel(DeformBwdArgs a) {
  const DeformColArgs& g = a.f;
  const long total = (long)g.B * g.Ho * g.Wo * g.KH * g.KW * g.C;
  const int cpg = g.C / g.DG;
  const bool wave_uniform = (g.C % 64 == 0) && (cpg % 64 == 0);
  for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
    const long idx = base + threadIdx.x;
    const bool live = idx < total;
    const long id = live ? idx : total - 1;
    const int c = (int)(id % g.C);
    long r = id / g.C;
    const int tap = (int)(r % (g.KH * g.KW)); r /= (g.KH * g.KW);
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int ho = (int)(r % g.Ho);
    const int b = (int)(r / g.Ho);
    const int dgi = c / cpg;
    const Taps t = deform_taps(g, b, ho, wo, tap / g.KW, tap % g.KW, dgi);
    const long row = ((long)b * g.Ho + ho) * g.Wo + wo;
    float gval = 0.f;
    if (live) gval = a.dcol_f32 ? ((const float*)a.dcol)[row * a.dcol_ld + (long)tap * g.C + c]
                                : bf2f(((const unsigned short*)a.dcol)[row * a.dcol_ld + (long)tap * g.C + c]);
    float doh = 0.f, dow = 0.f;
    if (live && t.inside) {
      const TIN* p = (const TIN*)g.data + (long)b * g.ds_b + (long)c * g.ds_c;
      const float v1 = dld<TIN>(p + (long)t.ya * g.ds_h + (long)t.xa * g.ds_w);
      const float v2 = dld<TIN>(p + (long)t.ya * g.ds_h + (long)t.xb * g.ds_w);
      const float v3 = dld<TIN>(p + (long)t.yb * g.ds_h + (long)t.xa * g.ds_w);
      const float v4 = dld<TIN>(p + (long)t.yb * g.ds_h + (long)t.xb * g.ds_w);
      // w1 = hh hw, w2 = hh lw, w3 = lh hw, w4 = lh lw with hh = 1 - lh, hw = 1 - lw; lh, lw move with the offsets
      // (a clamped axis has ya == yb / xa == xb, so its difference below vanishes like the reference's :199-210)
      const float hw = t.w1 + t.w3, lw = t.w2 + t.w4, hh = t.w1 + t.w2, lh = t.w3 + t.w4;
      doh = gval * (hw * (v3 - v1) + lw * (v4 - v2));
      dow = gval * (hh * (v2 - v1) + lh * (v4 - v3));
      float* gd = a.grad_data + (long)b * a.gs_b + (long)c * a.gs_c;
      atomicAdd(gd + (long)t.ya * a.gs_h + (long)t.xa * a.gs_w, t.w1 * gval);
      atomicAdd(gd + (long)t.ya * a.gs_h + (long)t.xb * a.gs_w, t.w2 * gval);
      atomicAdd(gd + (long)t.yb * a.gs_h + (long)t.xa * a.gs_w, t.w3 * gval);
      atomicAdd(gd + (long)t.yb * a.gs_h + (long)t.xb * a.gs_w, t.w4 * gval);
    }
    if (a.grad_offset) {
      float* go = a.grad_offset + (long)b * a.os_b + (long)ho * a.os_h + (long)wo * a.os_w +
                  (long)(dgi * 2 * g.KH * g.KW + 2 * tap) * a.os_c;
      if (wave_uniform) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { doh += __shfl_xor(doh, o); dow += __shfl_xor(dow, o); }
        if ((threadIdx.x & 63) == 0 && live) { atomicAdd(go, doh); atomicAdd(go + a.os_c, dow); }
      } else if (live && t.inside) {
        atomicAdd(go, doh); atomicAdd(go + a.os_c, dow);
      }
    }
  }
}

// (measured and dropped, r03: an LDS-aggregated form -- one workgroup per 8 x 16 pixel tile x 64 channels scattering into a
//  19 x 27-cell LDS window with ds_add_f32 and flushing it once -- cuts the memory-side atomics 8x but runs 3.8 ms (4 waves) /
//  2.4 ms (16 waves) against 1.24 ms of the kernel above at 8 images: with one 128 KiB workgroup per CU the serial
//  offset -> taps -> loads chain of each (pixel, tap) pair has nothing to hide behind, while the 64-channel-contiguous
//  global_atomic_add_f32 of the plain kernel coalesce into two 128-byte requests per wavefront.)

// ---- round 6: the data gradient as a GATHER ------------------------------------------------------------------------------------
// The scatter above issues four memory-side float atomics per (pixel, tap, channel); a feature cell of res5 receives ~36 of them
// (9 taps x 4 corner roles), and same-address atomics serialise: 353 M atomics = 1.24 ms per layer at 8 images, 13 % of the DCN
// training step.  The offsets only say WHERE a (pixel, tap) pair lands: a pair whose four (clamped) corners lie within D cells of its
// undeformed tap position (cy, cx) = (ho stride - pad + i dil, wo stride - pad + j dil) -- a NEAR pair, |offset| < D -- can only touch
// the (2 D + 1)^2 cells around (cy, cx).  So a cell (Y, X) collects, per tap, from the pairs whose (cy, cx) lie in its (2 D + 1)^2
// neighbourhood: 9 (2 D + 1)^2 candidate pairs (441 at the default D = 3), evaluated one per lane with exactly deform_taps() (same
// border rules, same rounding); then lane = channel(s) adds weight x dcol over the ~36 candidates that hit -- no atomics, one coalesced
// read-modify-write per cell.  FAR pairs (a corner further away: large learned offsets, clamped border taps) keep the atomic scatter in
// deformable_col2im_far_kernel, which tests the same predicate; the offset gradients come from deformable_col2im_offset_kernel.
// Stride 1, channels-last bf16 data / fp32 gradient, CPL = 1 or 2 channels per lane (64 CPL | channels per deformable group).
// Measured (tools/col2im_probe.py, profiles/r06_notes/col2im_gather.txt, res5 at 8 images, bf16 column gradient): scatter 1.23 ms; gather form with
// offsets of sigma 0.05 / 0.5 / 1.5 / 4 cells: D = 1 0.28 / 0.36 / 0.98 / 1.11 ms, D = 2 0.31 / 0.31 / 0.63 / 1.06 ms, D = 3 0.38 / 0.38 / 0.46 / 0.99 ms.
// D = 3 is the default: 0.07 ms more than D = 2 on untrained (near-zero) offsets, but trained offsets of a few cells stay on the fast path.
template <typename TCOL, int CPL>
__global__ __launch_bounds__(256) void deformable_col2im_gather_kernel(DeformBwdArgs a) {
  const DeformColArgs& g = a.f;
  constexpr int CW = 64 * CPL;                       // channels per wavefront
  const int lane = threadIdx.x & 63;
  const int chunks = g.C / CW, cpg = g.C / g.DG;
  const long nwave = (long)g.B * g.H * g.W * chunks;
  const long wid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wid >= nwave) return;
  const int ch = (int)(wid % chunks);
  long r = wid / chunks;
  const int X = (int)(r % g.W); r /= g.W;
  const int Y = (int)(r % g.H);
  const int b = (int)(r / g.H);
  const int dgi = (ch * CW) / cpg;
  const int D = a.win, side = 2 * D + 1, nwin = side * side;
  const int ntap = g.KH * g.KW, ncand = ntap * nwin;
  float acc[CPL];
#pragma unroll
  for (int e = 0; e < CPL; ++e) acc[e] = 0.f;
  const TCOL* dc = (const TCOL*)a.dcol + ch * CW + lane * CPL;
  for (int k0 = 0; k0 < ncand; k0 += 64) {
    // lane k evaluates candidate k: tap (i, j), undeformed tap position (Y + dy, X + dx)
    const int k = k0 + lane;
    float wgt = 0.f;
    int prow = 0, tap = 0;
    if (k < ncand) {
      tap = k / nwin;
      const int wv = k - tap * nwin, dy = wv / side - D, dx = wv - (wv / side) * side - D;
      const int i = tap / g.KW, j = tap - i * g.KW;
      const int cy = Y + dy, cx = X + dx;
      const int ho = cy + g.pad_h - i * g.dil_h, wo = cx + g.pad_w - j * g.dil_w;       // (stride 1)
      if (ho >= 0 && ho < g.Ho && wo >= 0 && wo < g.Wo) {
        const Taps t = deform_taps(g, b, ho, wo, i, j, dgi);
        if (t.inside && pair_near(t, cy, cx, D)) {
          if (t.ya == Y && t.xa == X) wgt += t.w1;
          if (t.ya == Y && t.xb == X) wgt += t.w2;
          if (t.yb == Y && t.xa == X) wgt += t.w3;
          if (t.yb == Y && t.xb == X) wgt += t.w4;
        }
        prow = (b * g.Ho + ho) * g.Wo + wo;
      }
    }
    unsigned long long hit = __ballot(wgt != 0.f);
    while (hit) {                    // four hits per trip: their loads are in flight together (a one-hit loop is one L2 round trip per hit)
      float w[4];
      const TCOL* src_p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool any = hit != 0ull;
        const int src = any ? __ffsll((long long)hit) - 1 : 0;
        hit &= hit - 1;              // (0 stays 0)
        w[u] = any ? __shfl(wgt, src, 64) : 0.f;
        const long row = __shfl(prow, src, 64);
        const int tp = __shfl(tap, src, 64);
        src_p[u] = dc + row * a.dcol_ld + (long)tp * g.C;        // (an exhausted slot re-reads lane 0's candidate row with weight 0: a valid address)
      }
      float v[4][CPL];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if constexpr (CPL == 1) v[u][0] = dld<TCOL>(src_p[u]);
        else if constexpr (sizeof(TCOL) == 4) { const float2 q = *(const float2*)src_p[u]; v[u][0] = q.x; v[u][1] = q.y; }
        else { const unsigned int q = *(const unsigned int*)src_p[u]; v[u][0] = __uint_as_float(q << 16); v[u][1] = __uint_as_float(q & 0xffff0000u); }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < CPL; ++e) acc[e] += w[u] * v[u][e];
    }
  }
  float* gd = a.grad_data + (long)b * a.gs_b + (long)Y * a.gs_h + (long)X * a.gs_w + (long)(ch * CW + lane * CPL);       // (gs_c == 1)
#pragma unroll
  for (int e = 0; e < CPL; ++e) gd[e] += acc[e];
}

// Offset gradients: one thread per (pixel, tap, 8-channel chunk) on channels-last bf16 data,
// four 16-byte corner loads; the two offset gradients are reduced over the chunk's 8 channels in registers, over the deformable
// group's chunks with lane shuffles (cpg / 8 consecutive lanes, a power of two <= 64), one atomic pair per group.
template <typename TCOL>
__global__ __launch_bounds__(256) void deformable_col2im_offset_kernel(DeformBwdArgs a) {
  const DeformColArgs& g = a.f;
  const int chunks = g.C / 8, cpg = g.C / g.DG, lpg = cpg / 8;      // lanes per deformable group
  const long total = (long)g.B * g.Ho * g.Wo * g.KH * g.KW * chunks;
  for (long base = (long)blockIdx.x * 256; base < total; base += (long)gridDim.x * 256) {
    const long idx = base + threadIdx.x;
    const bool live = idx < total;
    const long id = live ? idx : total - 1;
    const int cc = (int)(id % chunks);
    long r = id / chunks;
    const int tap = (int)(r % (g.KH * g.KW)); r /= (g.KH * g.KW);
    const int wo = (int)(r % g.Wo); r /= g.Wo;
    const int ho = (int)(r % g.Ho);
    const int b = (int)(r / g.Ho);
    const int c = cc * 8, dgi = c / cpg;
    const int i = tap / g.KW, j = tap - i * g.KW;
    const Taps t = deform_taps(g, b, ho, wo, i, j, dgi);
    const long row = ((long)b * g.Ho + ho) * g.Wo + wo;
    float doh = 0.f, dow = 0.f;
    if (live && t.inside) {
      float gv[8];
      const TCOL* dp = (const TCOL*)a.dcol + row * a.dcol_ld + (long)tap * g.C + c;
      if constexpr (sizeof(TCOL) == 4) {
        const float4 g0 = *(const float4*)dp, g1 = *(const float4*)(dp + 4);
        gv[0] = g0.x; gv[1] = g0.y; gv[2] = g0.z; gv[3] = g0.w; gv[4] = g1.x; gv[5] = g1.y; gv[6] = g1.z; gv[7] = g1.w;
      } else {
        const uint4 q = *(const uint4*)dp;
        const unsigned int u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { gv[2 * e] = __uint_as_float(u[e] << 16); gv[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u); }
      }
      const unsigned short* p = (const unsigned short*)g.data + (long)b * g.ds_b + c;
      const uint4 q1 = *(const uint4*)(p + (long)t.ya * g.ds_h + (long)t.xa * g.ds_w);
      const uint4 q2 = *(const uint4*)(p + (long)t.ya * g.ds_h + (long)t.xb * g.ds_w);
      const uint4 q3 = *(const uint4*)(p + (long)t.yb * g.ds_h + (long)t.xa * g.ds_w);
      const uint4 q4 = *(const uint4*)(p + (long)t.yb * g.ds_h + (long)t.xb * g.ds_w);
      const unsigned int* a1 = (const unsigned int*)&q1; const unsigned int* a2 = (const unsigned int*)&q2;
      const unsigned int* a3 = (const unsigned int*)&q3; const unsigned int* a4 = (const unsigned int*)&q4;
      const float hw = t.w1 + t.w3, lw = t.w2 + t.w4, hh = t.w1 + t.w2, lh = t.w3 + t.w4;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = e >> 1;
        const float v1 = (e & 1) ? __uint_as_float(a1[k] & 0xffff0000u) : __uint_as_float(a1[k] << 16);
        const float v2 = (e & 1) ? __uint_as_float(a2[k] & 0xffff0000u) : __uint_as_float(a2[k] << 16);
        const float v3 = (e & 1) ? __uint_as_float(a3[k] & 0xffff0000u) : __uint_as_float(a3[k] << 16);
        const float v4 = (e & 1) ? __uint_as_float(a4[k] & 0xffff0000u) : __uint_as_float(a4[k] << 16);
        doh += gv[e] * (hw * (v3 - v1) + lw * (v4 - v2));
        dow += gv[e] * (hh * (v2 - v1) + lh * (v4 - v3));
      }
    }
    if (a.grad_offset) {
      for (int o = lpg >> 1; o > 0; o >>= 1) { doh += __shfl_xor(doh, o, 64); dow += __shfl_xor(dow, o, 64); }
      if (live && (cc % lpg) == 0) {
        float* go = a.grad_offset + (long)b * a.os_b + (long)ho * a.os_h + (long)wo * a.os_w + (long)(dgi * 2 * g.KH * g.KW + 2 * tap) * a.os_c;
        atomicAdd(go, doh); atomicAdd(go + a.os_c, dow);
      }
    }
  }
}

// The FAR pairs' data gradient (companion of the two kernels above): lane = one (pixel, tap, deformable group) triple decides near / far with
// the same deform_taps() + pair_near(); the wavefront then walks the far ones together, lane = channel (coalesced column-gradient loads and
// float atomics).  While the learned offsets stay inside the window nothing is far and this is a pass over the offsets only.
template <typename TCOL>
__global__ __launch_bounds__(256) void deformable_col2im_far_kernel(DeformBwdArgs a) {
  const DeformColArgs& g = a.f;
  const int ntap = g.KH * g.KW, cpg = g.C / g.DG;
  const long total = (long)g.B * g.Ho * g.Wo * ntap * g.DG;
  const int lane = threadIdx.x & 63;
  for (long base = (long)blockIdx.x * 256 + (threadIdx.x & ~63); base < total; base += (long)gridDim.x * 256) {
    const long id = base + lane;
    bool far = false;
    Taps t{};
    int b = 0, dgi = 0, tap = 0;
    long row = 0;
    if (id < total) {
      dgi = (int)(id % g.DG);
      long r = id / g.DG;
      tap = (int)(r % ntap); r /= ntap;
      const int wo = (int)(r % g.Wo); r /= g.Wo;
      const int ho = (int)(r % g.Ho);
      b = (int)(r / g.Ho);
      const int i = tap / g.KW, j = tap - i * g.KW;
      t = deform_taps(g, b, ho, wo, i, j, dgi);
      const int cy = ho * g.stride_h - g.pad_h + i * g.dil_h, cx = wo * g.stride_w - g.pad_w + j * g.dil_w;
      far = t.inside && !pair_near(t, cy, cx, a.win);
      row = ((long)b * g.Ho + ho) * g.Wo + wo;
    }
    unsigned long long m = __ballot(far);
    while (m) {
      const int src = __ffsll((long long)m) - 1;
      m &= m - 1;
      const int ya = __shfl(t.ya, src, 64), yb = __shfl(t.yb, src, 64), xa = __shfl(t.xa, src, 64), xb = __shfl(t.xb, src, 64);
      const float w1 = __shfl(t.w1, src, 64), w2 = __shfl(t.w2, src, 64), w3 = __shfl(t.w3, src, 64), w4 = __shfl(t.w4, src, 64);
      const int sb = __shfl(b, src, 64), sd = __shfl(dgi, src, 64), st = __shfl(tap, src, 64);
      const long srow = __shfl(row, src, 64);
      for (int c = sd * cpg + lane; c < (sd + 1) * cpg; c += 64) {
        const float gv = dld<TCOL>((const TCOL*)a.dcol + srow * a.dcol_ld + (long)st * g.C + c);
        float* gd = a.grad_data + (long)sb * a.gs_b + (long)c * a.gs_c;
        atomicAdd(gd + (long)ya * a.gs_h + (long)xa * a.gs_w, w1 * gv);
        atomicAdd(gd + (long)ya * a.gs_h + (long)xb * a.gs_w, w2 * gv);
        atomicAdd(gd + (long)yb * a.gs_h + (long)xa * a.gs_w, w3 * gv);
        atomicAdd(gd + (long)yb * a.gs_h + (long)xb * a.gs_w, w4 * gv);
      }
    }
  }
}

struct PsroiBwdArgs {
  PsroiArgs f;                       // forward description (data values, rois, trans, shapes); f.out unused
  const void* grad_out; long go_r, go_c, go_ph, go_pw;   // [R, output_dim, P, P] (strides in elements, dtype of data)
  float* grad_data; long gs_b, gs_c, gs_h, gs_w;         // fp32 += (atomics)
  float* grad_trans;                                     // fp32 [R, 2*num_classes, part, part] += or nullptr
  int no_data;                                           // (c4 kernel) 1 = skip the data-gradient atomics: only grad_trans
};

// One axis of a bin's sample grid: the samples' bilinear corner weights summed per feature cell.  The reference adds
// (1-dx)(1-dy) diff, (1-dx) dy diff, ... per SAMPLE (deformable_psroi_pooling.cu:249-263); validity (:239-243), clamping and the
// corner weights are all per axis, so the spp x spp sample sum factors into (sum over valid iw of the x weights) x (sum over valid
// ih of the y weights): a bin touches n_x * n_y <= (2 spp)^2 distinct cells -- typically 2 x 2 or 3 x 3 -- instead of 4 spp^2
// atomics.  Entries are kept in registers (static indices, predicated updates).
struct AxisCells {
  int cell[8];
  float wt[8];
  int n, valid;
  __device__ __forceinline__ void add(int c, float w) {
    bool found = false;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < n && cell[k] == c) { wt[k] += w; found = true; }
    if (!found) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k == n) { cell[k] = c; wt[k] = w; }
      ++n;
    }
  }
  __device__ __forceinline__ void build(float start, float sub, int spp, float limit) {
    n = 0; valid = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) { cell[k] = 0; wt[k] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (i >= spp) break;
      float v = start + (float)i * sub;
      if (v < -0.5f || v > limit - 0.5f) continue;
      ++valid;
      v = fminf(fmaxf(v, 0.f), limit - 1.f);
      const int c0 = (int)floorf(v), c1 = (int)ceilf(v);
      const float d = v - (float)c0;
      add(c0, 1.f - d);
      if (c1 != c0) add(c1, d);
    }
  }
};

// Four channels per thread (round 6): with class-agnostic offsets (or none) and group_size 1 the roi geometry, the offset lookup and the per-axis cell sums
// of a bin are the same for every channel -- the kernel below rebuilds them per (roi, bin, channel), ~0.4 of its 1.29 ms.  Here a thread owns channels
// ctop0 + {0, 1, 2, 3} output_dim / 4 of one bin (a wavefront still covers 64 consecutive channels per step: coalesced atomics / loads).
template <typename T>
__global__ __launch_bounds__(256) void deformable_psroi_pool_bwd_c4_kernel(PsroiBwdArgs a) {
  const PsroiArgs& g = a.f;
  const int ocs = g.output_dim / 4;                       // channel stride between a thread's four channels
  const long total = (long)g.R * g.P * g.P * ocs;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int ctop0 = (int)(idx % ocs);
    long r = idx / ocs;
    const int pw = (int)(r % g.P); r /= g.P;
    const int ph = (int)(r % g.P);
    const int n = (int)(r / g.P);
    const RoiGeom q = psroi_geom(g, n);
    const int part_h = (int)floorf((float)ph / (float)g.P * (float)g.part);
    const int part_w = (int)floorf((float)pw / (float)g.P * (float)g.part);
    float tx = 0.f, ty = 0.f;
    long toff = 0;
    if (g.trans) {                                        // (one class: ch_each == output_dim)
      toff = (((long)n * 2) * g.part + part_h) * g.part + part_w;
      tx = g.trans[toff] * g.trans_std;
      ty = g.trans[toff + (long)g.part * g.part] * g.trans_std;
    }
    float wstart = (float)pw * q.bin_w + q.start_w; wstart = wstart + tx * q.roi_w;
    float hstart = (float)ph * q.bin_h + q.start_h; hstart = hstart + ty * q.roi_h;
    AxisCells ax, ay;
    ax.build(wstart, q.sub_w, g.spp, (float)g.W);
    ay.build(hstart, q.sub_h, g.spp, (float)g.H);
    const int count = ax.valid * ay.valid;
    float dtx = 0.f, dty = 0.f;
    if (count > 0) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        const int c = ctop0 + cc * ocs;                   // (group_size 1: input channel = output channel)
        const T* pc = (const T*)g.data + (long)q.b * g.ds_b + (long)c * g.ds_c;
        float* gd = a.grad_data + (long)q.b * a.gs_b + (long)c * a.gs_c;
        const float diff = dld<T>((const T*)a.grad_out + (long)n * a.go_r + (long)c * a.go_c + (long)ph * a.go_ph + (long)pw * a.go_pw) / (float)count;
        if (!a.no_data) {
#pragma unroll
          for (int ky = 0; ky < 8; ++ky) {
            if (ky >= ay.n) break;
            float* grow = gd + (long)ay.cell[ky] * a.gs_h;
            const float wy = ay.wt[ky] * diff;
#pragma unroll
            for (int kx = 0; kx < 8; ++kx) {
              if (kx >= ax.n) break;
              const float v = ax.wt[kx] * wy;
              if (v != 0.f) atomicAdd(grow + (long)ax.cell[kx] * a.gs_w, v);
            }
          }
        }
        if (a.grad_trans)
          for (int ih = 0; ih < g.spp; ++ih)
            for (int iw = 0; iw < g.spp; ++iw) {
              float w = wstart + (float)iw * q.sub_w, h = hstart + (float)ih * q.sub_h;
              if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) continue;
              w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
              h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
              const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
              const float dx = w - (float)x0, dy = h - (float)y0;
              const float u00 = dld<T>(pc + (long)y0 * g.ds_h + (long)x0 * g.ds_w), u01 = dld<T>(pc + (long)y1 * g.ds_h + (long)x0 * g.ds_w);
              const float u10 = dld<T>(pc + (long)y0 * g.ds_h + (long)x1 * g.ds_w), u11 = dld<T>(pc + (long)y1 * g.ds_h + (long)x1 * g.ds_w);
              dtx += (u11 * dy + u10 * (1.f - dy) - u01 * dy - u00 * (1.f - dy)) * g.trans_std * diff * q.roi_w;
              dty += (u11 * dx + u01 * (1.f - dx) - u10 * dx - u00 * (1.f - dx)) * g.trans_std * diff * q.roi_h;
            }
      }
    }
    if (a.grad_trans) {                                   // (ocs % 64 == 0: the whole wavefront shares the bin -> one atomic pair per wavefront)
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { dtx += __shfl_xor(dtx, o); dty += __shfl_xor(dty, o); }
      if ((threadIdx.x & 63) == 0 && (dtx != 0.f || dty != 0.f)) {
        atomicAdd(a.grad_trans + toff, dtx);
        atomicAdd(a.grad_trans + toff + (long)g.part * g.part, dty);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void deformable_psroi_pool_bwd_kernel(PsroiBwdArgs a) {
  const PsroiArgs& g = a.f;
  const long total = (long)g.R * g.P * g.P * g.output_dim;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int ctop = (int)(idx % g.output_dim);
    long r = idx / g.output_dim;
    const int pw = (int)(r % g.P); r /= g.P;
    const int ph = (int)(r % g.P);
    const int n = (int)(r / g.P);
    const RoiGeom q = psroi_geom(g, n);
    const int part_h = (int)floorf((float)ph / (float)g.P * (float)g.part);
    const int part_w = (int)floorf((float)pw / (float)g.P * (float)g.part);
    const int cls = ctop / g.ch_each;
    float tx = 0.f, ty = 0.f;
    long toff = 0;
    if (g.trans) {
      toff = ((((long)n * g.num_classes + cls) * 2) * g.part + part_h) * g.part + part_w;
      tx = g.trans[toff] * g.trans_std;
      ty = g.trans[toff + (long)g.part * g.part] * g.trans_std;
    }
    float wstart = (float)pw * q.bin_w + q.start_w; wstart = wstart + tx * q.roi_w;
    float hstart = (float)ph * q.bin_h + q.start_h; hstart = hstart + ty * q.roi_h;
    int gw = (int)floorf((float)pw * (float)g.group / (float)g.P);
    int gh = (int)floorf((float)ph * (float)g.group / (float)g.P);
    gw = min(max(gw, 0), g.group - 1); gh = min(max(gh, 0), g.group - 1);
    const int c = (ctop * g.group + gh) * g.group + gw;
    const T* pc = (const T*)g.data + (long)q.b * g.ds_b + (long)c * g.ds_c;
    float* gd = a.grad_data + (long)q.b * a.gs_b + (long)c * a.gs_c;
    const T* gop = (const T*)a.grad_out + (long)n * a.go_r + (long)ctop * a.go_c + (long)ph * a.go_ph + (long)pw * a.go_pw;
    if (g.spp <= 4) {
      AxisCells ax, ay;
      ax.build(wstart, q.sub_w, g.spp, (float)g.W);
      ay.build(hstart, q.sub_h, g.spp, (float)g.H);
      const int count = ax.valid * ay.valid;                  // the forward's top_count
      const float diff = count > 0 ? dld<T>(gop) / (float)count : 0.f;
      if (count > 0) {
#pragma unroll
        for (int ky = 0; ky < 8; ++ky) {
          if (ky >= ay.n) break;
          float* grow = gd + (long)ay.cell[ky] * a.gs_h;
          const float wy = ay.wt[ky] * diff;
#pragma unroll
          for (int kx = 0; kx < 8; ++kx) {
            if (kx >= ax.n) break;
            const float v = ax.wt[kx] * wy;
            if (v != 0.f) atomicAdd(grow + (long)ax.cell[kx] * a.gs_w, v);
          }
        }
      }
      if (a.grad_trans) {
        float dtx = 0.f, dty = 0.f;
        if (count > 0)
          for (int ih = 0; ih < g.spp; ++ih)
            for (int iw = 0; iw < g.spp; ++iw) {
              float w = wstart + (float)iw * q.sub_w, h = hstart + (float)ih * q.sub_h;
              if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) continue;
              w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
              h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
              const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
              const float dx = w - (float)x0, dy = h - (float)y0;
              const float u00 = dld<T>(pc + (long)y0 * g.ds_h + (long)x0 * g.ds_w), u01 = dld<T>(pc + (long)y1 * g.ds_h + (long)x0 * g.ds_w);
              const float u10 = dld<T>(pc + (long)y0 * g.ds_h + (long)x1 * g.ds_w), u11 = dld<T>(pc + (long)y1 * g.ds_h + (long)x1 * g.ds_w);
              dtx += (u11 * dy + u10 * (1.f - dy) - u01 * dy - u00 * (1.f - dy)) * g.trans_std * diff * q.roi_w;
              dty += (u11 * dx + u01 * (1.f - dx) - u10 * dx - u00 * (1.f - dx)) * g.trans_std * diff * q.roi_h;
            }
        // the channels of one class of a bin all add into the SAME two addresses (256 same-address atomics per bin with
        // class-agnostic offsets): when the whole wavefront shares the address, reduce over the lanes first
        const bool uniform = (g.output_dim % 64 == 0) && (g.ch_each % 64 == 0);
        if (uniform) {
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) { dtx += __shfl_xor(dtx, o); dty += __shfl_xor(dty, o); }
          if ((threadIdx.x & 63) == 0 && (dtx != 0.f || dty != 0.f)) {
            atomicAdd(a.grad_trans + toff, dtx);
            atomicAdd(a.grad_trans + toff + (long)g.part * g.part, dty);
          }
        } else if (count > 0) {
          atomicAdd(a.grad_trans + toff, dtx);
          atomicAdd(a.grad_trans + toff + (long)g.part * g.part, dty);
        }
      }
      continue;
    }
    // general sample counts: the reference's per-sample form
    int count = 0;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        const float w = wstart + (float)iw * q.sub_w, h = hstart + (float)ih * q.sub_h;
        if (!(w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f)) ++count;
      }
    if (count == 0) continue;
    const float diff = dld<T>(gop) / (float)count;
    float dtx = 0.f, dty = 0.f;
    for (int ih = 0; ih < g.spp; ++ih)
      for (int iw = 0; iw < g.spp; ++iw) {
        float w = wstart + (float)iw * q.sub_w, h = hstart + (float)ih * q.sub_h;
        if (w < -0.5f || w > (float)g.W - 0.5f || h < -0.5f || h > (float)g.H - 0.5f) continue;
        w = fminf(fmaxf(w, 0.f), (float)g.W - 1.f);
        h = fminf(fmaxf(h, 0.f), (float)g.H - 1.f);
        const int x0 = (int)floorf(w), x1 = (int)ceilf(w), y0 = (int)floorf(h), y1 = (int)ceilf(h);
        const float dx = w - (float)x0, dy = h - (float)y0;
        atomicAdd(gd + (long)y0 * a.gs_h + (long)x0 * a.gs_w, (1.f - dx) * (1.f - dy) * diff);
        atomicAdd(gd + (long)y1 * a.gs_h + (long)x0 * a.gs_w, (1.f - dx) * dy * diff);
        atomicAdd(gd + (long)y0 * a.gs_h + (long)x1 * a.gs_w, dx * (1.f - dy) * diff);
        atomicAdd(gd + (long)y1 * a.gs_h + (long)x1 * a.gs_w, dx * dy * diff);
        if (a.grad_trans) {
          const float u00 = dld<T>(pc + (long)y0 * g.ds_h + (long)x0 * g.ds_w), u01 = dld<T>(pc + (long)y1 * g.ds_h + (long)x0 * g.ds_w);
          const float u10 = dld<T>(pc + (long)y0 * g.ds_h + (long)x1 * g.ds_w), u11 = dld<T>(pc + (long)y1 * g.ds_h + (long)x1 * g.ds_w);
          dtx += (u11 * dy + u10 * (1.f - dy) - u01 * dy - u00 * (1.f - dy)) * g.trans_std * diff * q.roi_w;
          dty += (u11 * dx + u01 * (1.f - dx) - u10 * dx - u00 * (1.f - dx)) * g.trans_std * diff * q.roi_h;
        }
      }
    if (a.grad_trans) {
      atomicAdd(a.grad_trans + toff, dtx);
      atomicAdd(a.grad_trans + toff + (long)g.part * g.part, dty);
    }
  }
}

}  // namespace relnet

static int g_psroi_bwd_mode = 0;   // test / measurement knob: 0 auto, 1 = one channel per thread everywhere
extern "C" void relnet_deformable_psroi_pool_bwd_debug(int mode) { g_psroi_bwd_mode = mode; }

static int g_col2im_mode = 0;     // measurement / test knob: 0 auto (gather + offset + far-only kernels where they apply), 1 = the atomic scatter kernel, 2 = every pair treated as FAR, 10 + D = window radius D
extern "C" void relnet_deformable_col2im_debug(int mode) { g_col2im_mode = mode; }

extern "C" int relnet_deformable_col2im(const void* dcol, long dcol_ld, int dcol_dtype, const void* data,
                                        const long* data_strides4, int data_dtype, const float* offset,
                                        const long* offset_strides4, float* grad_data, const long* grad_data_strides4,
                                        float* grad_offset, const long* grad_offset_strides4, int B, int C, int H, int W,
                                        int KH, int KW, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                                        int dil_w, int num_deformable_group, void* stream) {
  RELNET_REQUIRE(dcol && data && offset && grad_data && data_strides4 && offset_strides4 && grad_data_strides4,
                 "relnet_deformable_col2im: null operand");
  RELNET_REQUIRE(!grad_offset || grad_offset_strides4, "relnet_deformable_col2im: grad_offset needs its strides");
  RELNET_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0 && KH > 0 && KW > 0 && num_deformable_group > 0 && C % num_deformable_group == 0,
                 "relnet_deformable_col2im: bad shape");
  DeformBwdArgs a;
  DeformColArgs& g = a.f;
  g.data = data; g.ds_b = data_strides4[0]; g.ds_c = data_strides4[1]; g.ds_h = data_strides4[2]; g.ds_w = data_strides4[3];
  g.offset = offset; g.fs_b = offset_strides4[0]; g.fs_c = offset_strides4[1]; g.fs_h = offset_strides4[2]; g.fs_w = offset_strides4[3];
  g.col = nullptr; g.col_ld = 0; g.B = B; g.C = C; g.H = H; g.W = W; g.KH = KH; g.KW = KW; g.pad_h = pad_h; g.pad_w = pad_w;
  g.stride_h = stride_h; g.stride_w = stride_w; g.dil_h = dil_h; g.dil_w = dil_w; g.DG = num_deformable_group;
  g.Ho = (H + 2 * pad_h - (dil_h * (KH - 1) + 1)) / stride_h + 1;
  g.Wo = (W + 2 * pad_w - (dil_w * (KW - 1) + 1)) / stride_w + 1;
  RELNET_REQUIRE(dcol_ld >= (long)KH * KW * C, "relnet_deformable_col2im: dcol_ld too small");
  a.dcol = dcol; a.dcol_ld = dcol_ld; a.dcol_f32 = dcol_dtype == RELNET_F32;
  a.grad_data = grad_data; a.gs_b = grad_data_strides4[0]; a.gs_c = grad_data_strides4[1]; a.gs_h = grad_data_strides4[2]; a.gs_w = grad_data_strides4[3];
  a.grad_offset = grad_offset;
  if (grad_offset) { a.os_b = grad_offset_strides4[0]; a.os_c = grad_offset_strides4[1]; a.os_h = grad_offset_strides4[2]; a.os_w = grad_offset_strides4[3]; }
  hipStream_t s = (hipStream_t)stream;
  // gather form (round 6): stride 1, channels-last bf16 data, dense channels-last fp32 gradient, whole 64-channel runs per deformable group,
  // lanes-per-group a power of two, 16-byte aligned rows.  g_col2im_mode: 0 auto (window radius 3), 1 = scatter kernel only, 2 = every pair
  // declared FAR (offset kernel + the scatter kernel's far-only pass: the consistency check of the two), 10 + D = window radius D
  a.win = 0;
  const int cpg = C / num_deformable_group, lpg = cpg / 8;
  const bool gather_ok = g_col2im_mode != 1 && data_dtype == RELNET_BF16 && stride_h == 1 && stride_w == 1 && g.Ho == H && g.Wo == W &&
                         g.ds_c == 1 && a.gs_c == 1 && cpg % 64 == 0 && lpg <= 64 && (lpg & (lpg - 1)) == 0 && (64 % lpg) == 0 &&
                         g.ds_h % 8 == 0 && g.ds_w % 8 == 0 && g.ds_b % 8 == 0 && (((uintptr_t)data) & 15) == 0 &&
                         dcol_ld % 8 == 0 && (((uintptr_t)dcol) & 15) == 0 && (long)B * g.Ho * g.Wo < (1L << 31);
  if (gather_ok) {
    const bool all_far = g_col2im_mode == 2;
    a.win = all_far ? -1 : (g_col2im_mode >= 10 ? g_col2im_mode - 10 : 3);       // (-1: no pair is near)
    const int cpl = (cpg % 128 == 0) ? 2 : 1;
    const long nwave = (long)B * H * W * (C / (64 * cpl));
    const unsigned ggrid = (unsigned)((nwave + 3) / 4), ogrid = grid_for((long)B * g.Ho * g.Wo * KH * KW * (C / 8));
    if (a.dcol_f32) {
      if (!all_far) { if (cpl == 2) deformable_col2im_gather_kernel<float, 2><<<ggrid, 256, 0, s>>>(a); else deformable_col2im_gather_kernel<float, 1><<<ggrid, 256, 0, s>>>(a); }
      deformable_col2im_offset_kernel<float><<<ogrid, 256, 0, s>>>(a);
    } else {
      if (!all_far) { if (cpl == 2) deformable_col2im_gather_kernel<unsigned short, 2><<<ggrid, 256, 0, s>>>(a); else deformable_col2im_gather_kernel<unsigned short, 1><<<ggrid, 256, 0, s>>>(a); }
      deformable_col2im_offset_kernel<unsigned short><<<ogrid, 256, 0, s>>>(a);
    }
    // the pairs beyond the window: atomic scatter (a pass over the offsets only while they stay below the window radius)
    const unsigned fgrid = grid_for((long)B * g.Ho * g.Wo * KH * KW * num_deformable_group);
    if (a.dcol_f32) deformable_col2im_far_kernel<float><<<fgrid, 256, 0, s>>>(a);
    else deformable_col2im_far_kernel<unsigned short><<<fgrid, 256, 0, s>>>(a);
    return check_launch("relnet_deformable_col2im");
  }
  const unsigned grid = grid_for((long)B * g.Ho * g.Wo * KH * KW * C);
  if (data_dtype == RELNET_F32) deformable_col2im_kernel<float><<<grid, 256, 0, s>>>(a);
  else if (data_dtype == RELNET_BF16) deformable_col2im_kernel<unsigned short><<<grid, 256, 0, s>>>(a);
  else RELNET_REQUIRE(false, "relnet_deformable_col2im: unknown dtype %d", data_dtype);
  return check_launch("relnet_deformable_col2im");
}

extern "C" int relnet_deformable_psroi_pool_bwd(const void* grad_out, const long* grad_out_strides4, const void* data,
                                                const long* data_strides4, const float* rois, const float* trans,
                                                float* grad_data, const long* grad_data_strides4, float* grad_trans,
                                                int R, int C, int H, int W, int output_dim, int group_size,
                                                int pooled_size, int part_size, int sample_per_part, float spatial_scale,
                                                float trans_std, int num_classes, int batch_index_base, int dtype,
                                                void* stream) {
  RELNET_REQUIRE(grad_out && grad_out_strides4 && data && data_strides4 && rois && grad_data && grad_data_strides4,
                 "relnet_deformable_psroi_pool_bwd: null operand");
  RELNET_REQUIRE(R > 0 && C == output_dim * group_size * group_size && pooled_size > 0 && sample_per_part > 0,
                 "relnet_deformable_psroi_pool_bwd: bad shape");
  const bool no_trans = (trans == nullptr);
  RELNET_REQUIRE(no_trans || (num_classes > 0 && output_dim % num_classes == 0), "relnet_deformable_psroi_pool_bwd: bad num_classes");
  RELNET_REQUIRE(no_trans || grad_trans, "relnet_deformable_psroi_pool_bwd: grad_trans required when trans is given");
  PsroiBwdArgs a;
  PsroiArgs& g = a.f;
  g.data = data; g.ds_b = data_strides4[0]; g.ds_c = data_strides4[1]; g.ds_h = data_strides4[2]; g.ds_w = data_strides4[3];
  g.rois = rois; g.trans = trans; g.out = nullptr; g.os_r = g.os_c = g.os_ph = g.os_pw = 0; g.top_count = nullptr;
  g.R = R; g.H = H; g.W = W; g.output_dim = output_dim; g.group = group_size; g.P = pooled_size;
  g.part = part_size > 0 ? part_size : pooled_size; g.spp = sample_per_part;
  g.num_classes = no_trans ? 1 : num_classes; g.ch_each = no_trans ? output_dim : output_dim / num_classes;
  g.batch_index_base = batch_index_base; g.scale = spatial_scale; g.trans_std = trans_std;
  a.grad_out = grad_out; a.go_r = grad_out_strides4[0]; a.go_c = grad_out_strides4[1]; a.go_ph = grad_out_strides4[2]; a.go_pw = grad_out_strides4[3];
  a.grad_data = grad_data; a.gs_b = grad_data_strides4[0]; a.gs_c = grad_data_strides4[1]; a.gs_h = grad_data_strides4[2]; a.gs_w = grad_data_strides4[3];
  a.grad_trans = no_trans ? nullptr : grad_trans;
  a.no_data = 0;
  const unsigned grid = grid_for((long)R * pooled_size * pooled_size * output_dim);
  hipStream_t s = (hipStream_t)stream;
  // four channels per thread where the per-bin work is channel independent (g_psroi_bwd_mode 1 = the one-channel kernel)
  if (g_psroi_bwd_mode != 1 && sample_per_part <= 4 && group_size == 1 && output_dim == C && g.num_classes == 1 && output_dim % 256 == 0 &&
      (dtype == RELNET_F32 || dtype == RELNET_BF16)) {
    const unsigned grid4 = grid_for((long)R * pooled_size * pooled_size * (output_dim / 4));
    if (dtype == RELNET_F32) deformable_psroi_pool_bwd_c4_kernel<float><<<grid4, 256, 0, s>>>(a);
    else deformable_psroi_pool_bwd_c4_kernel<unsigned short><<<grid4, 256, 0, s>>>(a);
    return check_launch("relnet_deformable_psroi_pool_bwd");
  }
  if (dtype == RELNET_F32) deformable_psroi_pool_bwd_kernel<float><<<grid, 256, 0, s>>>(a);
  else if (dtype == RELNET_BF16) deformable_psroi_pool_bwd_kernel<unsigned short><<<grid, 256, 0, s>>>(a);
  else RELNET_REQUIRE(false, "relnet_deformable_psroi_pool_bwd: unknown dtype %d", dtype);
  return check_launch("relnet_deformable_psroi_pool_bwd");
}
