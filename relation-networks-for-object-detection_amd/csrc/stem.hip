// Stem epilogue: bias + ReLU + 3x3/2 max-pool (pooling_convention='full' = ceil mode) in one pass
// over the NHWC conv1 output (reference: bn_conv1 -> conv1_relu -> pool1,
// relation_rcnn/symbols/resnet_v1_101_rcnn_base.py:30-36).  Adding the folded-BN bias and the ReLU
// commute with the max (both monotonic), so they are applied once per pooled element: the library
// path spends three full read+write passes over the 64x300x500 map per image on them.
#include "common.h"

namespace relnet {

struct StemArgs {
  const unsigned short* in;   // [B, H, W, C] bf16 (conv output WITHOUT bias)
  const float* bias;          // [C]
  unsigned short* out;        // [B, Ho, Wo, C] bf16
  int B, H, W, C, Ho, Wo, ksize, stride;
};

__global__ __launch_bounds__(256) void stem_bias_relu_pool_kernel(StemArgs g) {
  const int groups = g.C >> 3;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)g.B * g.Ho * g.Wo * groups;
  if (t >= total) return;
  const int cg = (int)(t % groups);
  long p = t / groups;
  const int ox = (int)(p % g.Wo); p /= g.Wo;
  const int oy = (int)(p % g.Ho);
  const int b = (int)(p / g.Ho);
  const int y0 = oy * g.stride, x0 = ox * g.stride;
  const int y1 = min(y0 + g.ksize, g.H), x1 = min(x0 + g.ksize, g.W);
  float best[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
  const unsigned short* base = g.in + (long)b * g.H * g.W * g.C + cg * 8;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const uint4 v = *(const uint4*)(base + ((long)y * g.W + x) * g.C);
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        best[2 * e] = fmaxf(best[2 * e], bf2f(w4[e] & 0xffff));
        best[2 * e + 1] = fmaxf(best[2 * e + 1], bf2f(w4[e] >> 16));
      }
    }
  const float4 b0 = *(const float4*)(g.bias + cg * 8), b1 = *(const float4*)(g.bias + cg * 8 + 4);
  const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
  float r[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = fmaxf(bf2f(f2bf(best[e] + bb[e])), 0.f);   // round like the separate bias pass
  *(uint4*)(g.out + (((long)b * g.Ho + oy) * g.Wo + ox) * g.C + cg * 8) =
      make_uint4(pack_bf16x2(r[0], r[1]), pack_bf16x2(r[2], r[3]), pack_bf16x2(r[4], r[5]), pack_bf16x2(r[6], r[7]));
}

}  // namespace relnet

using namespace relnet;

// in [B,H,W,C] bf16 NHWC (C % 8 == 0), out [B,Ho,Wo,C] with Ho = ceil((H - k) / s) + 1 (ceil mode,
// windows clipped at the border, no padding).
extern "C" int relnet_stem_bias_relu_pool(const void* in, const float* bias, void* out, int B, int H, int W,
                                          int C, int ksize, int stride, void* stream) {
  RELNET_REQUIRE(in && bias && out, "relnet_stem_bias_relu_pool: null operand");
  RELNET_REQUIRE(B > 0 && H >= ksize && W >= ksize && C > 0 && C % 8 == 0 && stride > 0, "relnet_stem_bias_relu_pool: bad shape");
  StemArgs g;
  g.in = (const unsigned short*)in; g.bias = bias; g.out = (unsigned short*)out;
  g.B = B; g.H = H; g.W = W; g.C = C; g.ksize = ksize; g.stride = stride;
  g.Ho = (H - ksize + stride - 1) / stride + 1;
  g.Wo = (W - ksize + stride - 1) / stride + 1;
  if ((g.Ho - 1) * stride >= H) --g.Ho;      // last window must start inside the input
  if ((g.Wo - 1) * stride >= W) --g.Wo;
  const long total = (long)B * g.Ho * g.Wo * (C / 8);
  stem_bias_relu_pool_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_stem_bias_relu_pool");
}
