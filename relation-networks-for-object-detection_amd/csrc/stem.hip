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

// pool1 of the float32 parity path (mx.symbol.Pooling(kernel 3x3, stride 2, pool_type 'max', pooling_convention 'full'),
// resnet_v1_101_rcnn_base.py:35-36): ceil-mode windows clipped at the border, no padding.  in [B,H,W,C] fp32 NHWC, C % 4 == 0.
namespace relnet {
struct PoolF32Args { const float* in; float* out; int B, H, W, C, Ho, Wo, ksize, stride; };
__global__ __launch_bounds__(256) void maxpool_nhwc_f32_kernel(PoolF32Args g) {
  const int groups = g.C >> 2;
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= (long)g.B * g.Ho * g.Wo * groups) return;
  const int cg = (int)(t % groups);
  long p = t / groups;
  const int ox = (int)(p % g.Wo); p /= g.Wo;
  const int oy = (int)(p % g.Ho);
  const int b = (int)(p / g.Ho);
  const int y0 = oy * g.stride, x0 = ox * g.stride;
  const int y1 = min(y0 + g.ksize, g.H), x1 = min(x0 + g.ksize, g.W);
  float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
  const float* base = g.in + (long)b * g.H * g.W * g.C + cg * 4;
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const float4 v = *(const float4*)(base + ((long)y * g.W + x) * g.C);
      best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
    }
  *(float4*)(g.out + (((long)b * g.Ho + oy) * g.Wo + ox) * g.C + cg * 4) = best;
}
}  // namespace relnet

extern "C" int relnet_maxpool_nhwc_f32(const float* in, float* out, int B, int H, int W, int C, int ksize, int stride, void* stream) {
  RELNET_REQUIRE(in && out, "relnet_maxpool_nhwc_f32: null operand");
  RELNET_REQUIRE(B > 0 && H >= ksize && W >= ksize && C > 0 && C % 4 == 0 && stride > 0, "relnet_maxpool_nhwc_f32: bad shape");
  relnet::PoolF32Args g{in, out, B, H, W, C, 0, 0, ksize, stride};
  g.Ho = (H - ksize + stride - 1) / stride + 1;
  g.Wo = (W - ksize + stride - 1) / stride + 1;
  if ((g.Ho - 1) * stride >= H) --g.Ho;      // last window must start inside the input
  if ((g.Wo - 1) * stride >= W) --g.Wo;
  const long total = (long)B * g.Ho * g.Wo * (C / 4);
  relnet::maxpool_nhwc_f32_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(g);
  return relnet::check_launch("relnet_maxpool_nhwc_f32");
}

// ---------------------------------------------------------------------------------------
// Fused stem: conv1 7x7 / 2 (pad 3, Cin = 3) + folded-BN bias + ReLU + pool1 3x3 / 2 (ceil mode) in ONE kernel, from the
// raw NCHW image to the pooled NHWC bf16 map (reference graph: resnet_v1_101_rcnn_base.py:30-36).  The three-launch path
// (relnet_stem_pack_input, relnet_stem_conv7, relnet_stem_bias_relu_pool) writes the 64 x 300 x 500 conv map to HBM
// (1.04 GB at 54 images) and reads it back: 1.07 ms of a 22.4 ms step for 1 % of its FLOPs.  Here a workgroup owns a tile of
// 8 x 14 POOLED pixels:
//   * the 39 x 63 input window (17 x 29 conv pixels x stride 2 + 5) is converted to zero-padded NHWC4 bf16 in LDS (20 KB);
//     K = 7 tap rows x 8 taps x 4 channels = 224 (tap 7 and channel 3 carry zero weights): the 8 k-values of an MFMA
//     fragment are two neighbouring input pixels = 16 contiguous LDS bytes, so the pixel fragments are read straight from the
//     window (no im2col image at all) and neighbouring lanes (conv pixels) read neighbouring 16-byte chunks;
//   * the weights sit in LDS in fragment order (28 KB); D = W x A (lane <-> conv pixel, registers <-> channels) like the
//     other convolution kernels, same k order as the three-launch path: bit-identical results;
//   * bias + ReLU + bf16, conv tile -> LDS (64 KB, aliasing window + weights), 3 x 3 max over it, 16-byte stores.
// ---------------------------------------------------------------------------------------
namespace relnet {

constexpr int kSfPY = 8, kSfPX = 14;                       // pooled tile
constexpr int kSfCY = 2 * kSfPY + 1, kSfCX = 2 * kSfPX + 1; // conv tile 17 x 29 = 493 pixels (16 MFMA row tiles)
constexpr int kSfIY = 2 * kSfCY + 5, kSfIX = 64;           // input window 39 rows x 63 (+1) columns
constexpr int kSfKS = 14;                                  // k-steps of 16: (tap row 0..6) x (tap pair 0, 1)
constexpr int kSfOutLd = 64 + 8;                           // conv tile row pitch in LDS (elements): 144 B

struct StemFusedArgs {
  const void* in; int in_bf16;        // [B, 3, H, W] NCHW fp32 / bf16
  const unsigned short* w256;         // [64][256] bf16, k = ty * 32 + tx * 4 + c (ops.pack_stem_weight)
  const float* bias;                  // [64]
  unsigned short* out;                // [B, Hp, Wp, 64] bf16
  int B, H, W, Hc, Wc, Hp, Wp, tiles_y, tiles_x;
};

__global__ __launch_bounds__(512) void stem_fused_kernel(StemFusedArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* sIn = (unsigned short*)smem;                                   // [39][64][4]
  unsigned short* sW = (unsigned short*)(smem + kSfIY * kSfIX * 8);              // [14][2][64 lanes][8]
  unsigned short* sOut = (unsigned short*)smem;                                  // [512][kSfOutLd] (after the MFMAs)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  int t = blockIdx.x;
  const int tx = t % g.tiles_x; t /= g.tiles_x;
  const int ty = t % g.tiles_y;
  const int b = t / g.tiles_y;
  const int cy0 = ty * 2 * kSfPY, cx0 = tx * 2 * kSfPX;          // first conv pixel of the tile
  const int iy0 = 2 * cy0 - 3, ix0 = 2 * cx0 - 3;                // first input pixel of the window (pad 3)

  // ---- weights -> LDS in fragment order: frag (kk, nt), lane l <- W[32 nt + (l & 31)][ty * 32 + (kk & 1) * 16 + 8 (l >> 5) ..]
  // ---- input window -> zero-padded NHWC4 bf16.
  // Every global load of the prologue (4 weight chunks + 15 pixel values per thread) is ISSUED before the first one is consumed: with two
  // 8-wave workgroups per CU nothing else hides the latency, and the rolled loops (load -> wait -> LDS store, five times for the window)
  // cost a round trip each (r06, 108 images: 0.98 -> 0.81 ms per launch).  Out-of-range pixels load a
  // clamped address and are zeroed afterwards, so no load sits behind a branch.
  constexpr int kWIt = (kSfKS * 2 * 64 + 511) / 512;             // 4 (the last one half full)
  constexpr int kPIt = (kSfIY * kSfIX + 511) / 512;              // 5
  uint4 wv[kWIt];
#pragma unroll
  for (int it = 0; it < kWIt; ++it) {
    const int c = min(tid + 512 * it, kSfKS * 2 * 64 - 1);
    const int l = c & 63, nt = (c >> 6) & 1, kk = c >> 7;
    const int k = (kk >> 1) * 32 + (kk & 1) * 16 + 8 * (l >> 5);
    wv[it] = *(const uint4*)(g.w256 + (32 * nt + (l & 31)) * 256 + k);
  }
  const long plane = (long)g.H * g.W;
  float pv[kPIt][3];
  bool pok[kPIt];
#pragma unroll
  for (int it = 0; it < kPIt; ++it) {
    const int p = tid + 512 * it;
    const int wy = p / kSfIX, wx = p - wy * kSfIX;
    const int y = iy0 + wy, x = ix0 + wx;
    pok[it] = p < kSfIY * kSfIX && wx < kSfIX - 1 && y >= 0 && y < g.H && x >= 0 && x < g.W;
    const long o = (long)b * 3 * plane + (long)min(max(y, 0), g.H - 1) * g.W + min(max(x, 0), g.W - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c)
      pv[it][c] = g.in_bf16 ? bf2f(((const unsigned short*)g.in)[o + c * plane]) : ((const float*)g.in)[o + c * plane];
  }
#pragma unroll
  for (int it = 0; it < kWIt; ++it) {
    const int c = tid + 512 * it;
    if (c < kSfKS * 2 * 64) *(uint4*)(sW + (long)c * 8) = wv[it];
  }
#pragma unroll
  for (int it = 0; it < kPIt; ++it) {
    const int p = tid + 512 * it;
    if (p < kSfIY * kSfIX)
      *(uint2*)(sIn + (long)p * 4) = pok[it] ? make_uint2(pack_bf16x2(pv[it][0], pv[it][1]), pack_bf16x2(pv[it][2], 0.f)) : make_uint2(0u, 0u);
  }
  __syncthreads();

  // ---- MFMAs: wave w owns conv pixels [64 w, 64 w + 64) (2 column tiles of 32 pixels) x 64 channels (2 row tiles)
  f32x16 acc[2][2];                                              // [channel tile][pixel tile]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  int pbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int m = wave * 64 + j * 32 + l31;
    m = m < kSfCY * kSfCX ? m : 0;                               // rows past the tile: any valid pixel (never stored)
    const int cy = m / kSfCX, cx = m - cy * kSfCX;
    pbase[j] = ((2 * cy) * kSfIX + 2 * cx + 2 * half) * 4;       // element offset of tap (0, 2 half) of this conv pixel
  }
#pragma unroll
  for (int kk = 0; kk < kSfKS; ++kk) {
    const int tyy = kk >> 1, txo = (kk & 1) * 4;
    bf16x8 wf[2], pf[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) wf[i] = *(const bf16x8*)(sW + ((kk * 2 + i) * 64 + lane) * 8);
#pragma unroll
    for (int j = 0; j < 2; ++j) pf[j] = *(const bf16x8*)(sIn + pbase[j] + (tyy * kSfIX + txo) * 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], pf[j], acc[i][j], 0, 0, 0);
  }
  __syncthreads();                                               // window + weights are dead: the conv tile takes their place

  // ---- bias + ReLU + bf16 -> conv tile in LDS.  acc[i][j][r]: channel 32 i + (r & 3) + 8 (r >> 2) + 4 half, pixel = this lane's
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = wave * 64 + j * 32 + l31;
    const int cy = m / kSfCX, cx = m - cy * kSfCX;
    const bool live = m < kSfCY * kSfCX && cy0 + cy < g.Hc && cx0 + cx < g.Wc;      // conv pixels outside the map pool as 0
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int ch = 32 * i + 8 * gq + 4 * half;
        const float4 bv = *(const float4*)(g.bias + ch);
        float v0 = fmaxf(acc[i][j][4 * gq] + bv.x, 0.f), v1 = fmaxf(acc[i][j][4 * gq + 1] + bv.y, 0.f);
        float v2 = fmaxf(acc[i][j][4 * gq + 2] + bv.z, 0.f), v3 = fmaxf(acc[i][j][4 * gq + 3] + bv.w, 0.f);
        if (!live) { v0 = v1 = v2 = v3 = 0.f; }
        *(uint2*)(sOut + (long)m * kSfOutLd + ch) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
  }
  __syncthreads();

  // ---- pool1: 3 x 3 / 2 max over the conv tile (all values >= 0: pixels outside the map were stored as 0), 16-byte stores
  for (int it = tid; it < kSfPY * kSfPX * 8; it += 512) {
    const int cg = it & 7, pp = it >> 3;
    const int py = pp / kSfPX, px = pp - py * kSfPX;
    const int oy = ty * kSfPY + py, ox = tx * kSfPX + px;
    if (oy >= g.Hp || ox >= g.Wp) continue;
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const uint4 v = *(const uint4*)(sOut + (long)((2 * py + dy) * kSfCX + 2 * px + dx) * kSfOutLd + cg * 8);
        const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          best[2 * e] = fmaxf(best[2 * e], bf2f(w4[e] & 0xffff));
          best[2 * e + 1] = fmaxf(best[2 * e + 1], bf2f(w4[e] >> 16));
        }
      }
    *(uint4*)(g.out + (((long)b * g.Hp + oy) * g.Wp + ox) * 64 + cg * 8) =
        make_uint4(pack_bf16x2(best[0], best[1]), pack_bf16x2(best[2], best[3]), pack_bf16x2(best[4], best[5]), pack_bf16x2(best[6], best[7]));
  }
}

}  // namespace relnet

// data [B,3,H,W] NCHW (fp32: in_dtype 0, bf16: 1), w256 = ops.pack_stem_weight image [64][256] bf16, bias [64] fp32
// -> out [B,Hp,Wp,64] bf16 NHWC with Hc = (H + 6 - 7) / 2 + 1, Hp = ceil((Hc - 3) / 2) + 1 (last window starts inside).
extern "C" int relnet_stem_fused(const void* data, int in_dtype, const void* w256, const float* bias, void* out, int B,
                                 int H, int W, void* stream) {
  RELNET_REQUIRE(data && w256 && bias && out, "relnet_stem_fused: null operand");
  RELNET_REQUIRE(B > 0 && H >= 7 && W >= 7, "relnet_stem_fused: bad shape");
  RELNET_REQUIRE(((uintptr_t)w256 & 15) == 0 && ((uintptr_t)out & 15) == 0 && ((uintptr_t)bias & 15) == 0, "relnet_stem_fused: operands must be 16-byte aligned");
  StemFusedArgs g;
  g.in = data; g.in_bf16 = in_dtype == 1; g.w256 = (const unsigned short*)w256; g.bias = bias; g.out = (unsigned short*)out;
  g.B = B; g.H = H; g.W = W;
  g.Hc = (H + 6 - 7) / 2 + 1; g.Wc = (W + 6 - 7) / 2 + 1;
  g.Hp = (g.Hc - 3 + 1) / 2 + 1; g.Wp = (g.Wc - 3 + 1) / 2 + 1;
  if ((g.Hp - 1) * 2 >= g.Hc) --g.Hp;
  if ((g.Wp - 1) * 2 >= g.Wc) --g.Wp;
  g.tiles_y = (g.Hp + kSfPY - 1) / kSfPY; g.tiles_x = (g.Wp + kSfPX - 1) / kSfPX;
  const size_t lds = (size_t)512 * kSfOutLd * 2;            // 73 728 B: conv tile (>= window 19 968 + weights 28 672)
  static relnet::PerDeviceOnce attr_once;
  if (attr_once.first()) hipFuncSetAttribute((const void*)stem_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  stem_fused_kernel<<<(unsigned)((long)B * g.tiles_y * g.tiles_x), 512, lds, (hipStream_t)stream>>>(g);
  return check_launch("relnet_stem_fused");
}
