// Pixel-wise chain across a residual-block boundary of the ResNet trunk (reference graph:
// relation_rcnn/symbols/resnet_v1_101_rcnn_base.py, e.g. res2a_branch2c .. res2b_branch2a, :66-84):
//
//     x_next = relu(W3 . mid2 + b3 + x)            1x1 expand (BN folded) + identity shortcut + ReLU of block n
//     mid1'  = relu(W1' . x_next + b1')            1x1 reduce (BN folded) + ReLU of block n+1
//
// Both are 1x1 convolutions, i.e. per-pixel products, so block n+1's reduce can consume x_next while it is still in
// registers: the 256- (512-) channel activation is written once and NOT read back by a separate reduce kernel.  For the
// HBM-bound stages that is the dominant saving (res2 at B = 54: expand 535 us + reduce 256 us as two GEMM launches;
// DESIGN.md section 4 has the measured figure of this kernel).
//
// One wavefront = 32 pixels (independent of every other wavefront after the weights are in LDS):
//   phase A  acc^T[cout][px] = W3 (A operand, rows = cout) x mid2^T (B operand: lane = pixel, 8 channels per k-step, read
//            straight from HBM: a pixel row of mid2 is one 128-byte line), 128 output channels per pass;
//   shortcut the 32 x 128-channel slice of x arrives in LDS by global_load_lds (1 KiB per instruction = 4 pixel rows,
//            16-byte chunks XOR-swizzled on the SOURCE side so that the per-lane 8-byte reads spread over the banks);
//            relu(acc + b3 + x) is written back IN PLACE as bf16 and leaves as 16-byte coalesced rows;
//   phase B  the same packed bf16 values are the B operand of the second product (the contraction index cout is permuted
//            identically in the accumulator registers and in the pre-packed W1' fragments, as in the attention kernel),
//            mid1'^T[c][px] accumulates over the two passes, then bias + ReLU + LDS transpose + coalesced rows.
// Rounding points are those of the two-launch path (x_next is rounded to bf16 before the reduce product).
#include "common.h"

namespace relnet {

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

struct ChainArgs {
  const unsigned short* m2;   // [P][MID]   3x3 output of block n (bf16)
  const unsigned short* x;    // [P][4 MID] shortcut = input of block n
  const uint4* w3f;           // W3 [4 MID][MID] in fragment order (relnet_pack_w_frag)
  const uint4* w1f;           // W1' [MID][4 MID], fragment order with the accumulator permutation (ops.pack_chain_w1)
  const float* b3;            // [4 MID]
  const float* b1;            // [MID]
  unsigned short* xn;         // [P][4 MID]
  unsigned short* m1;         // [P][MID]
  int P;
};

// MID = 64 (res2): W3 (32 KiB) and W1' (32 KiB) stay in LDS for the life of the workgroup; 8 KiB stage per wavefront.
__global__ __launch_bounds__(512) void bottleneck_chain64_kernel(ChainArgs a) {
  constexpr int MID = 64, COUT = 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint4* sW3 = (uint4*)smem;                                  // [8 tiles][4 ks][64 lanes]
  uint4* sW1 = sW3 + 8 * 4 * 64;                              // [2 tiles][16 ks][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  unsigned char* stage = smem + 65536 + wave * 8192;          // [32 px][256 B], chunk c of row r at position c ^ (r & 15)
  for (int i = tid; i < 2048; i += 512) sW3[i] = a.w3f[i];
  for (int i = tid; i < 2048; i += 512) sW1[i] = a.w1f[i];
  __syncthreads();
  const int ntile = (a.P + 31) / 32;
  const int drow = lane >> 4, dcp = lane & 15;                // DMA / coalesced-store role of this lane: row 4 i + drow, chunk slot dcp
  for (int tile = blockIdx.x * 8 + wave; tile < ntile; tile += gridDim.x * 8) {
    const int p0 = tile * 32;
    const int px = min(p0 + l31, a.P - 1);
    bf16x8 m2f[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) m2f[ks] = *(const bf16x8*)(a.m2 + (long)px * MID + 16 * ks + 8 * half);
    f32x16 m1acc[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m1acc[rt][r] = 0.f;
#pragma unroll 1
    for (int hf = 0; hf < 2; ++hf) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // LDS reads of the previous pass are done before the DMA overwrites
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + drow;
        const unsigned short* src = a.x + (long)min(p0 + row, a.P - 1) * COUT + hf * 128 + ((dcp ^ (row & 15)) << 3);
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(stage + i * 1024), 16, 0, 0);
      }
      f32x16 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float4 bv = *(const float4*)(a.b3 + hf * 128 + ct * 32 + 8 * g + 4 * half);
          acc[ct][4 * g] = bv.x; acc[ct][4 * g + 1] = bv.y; acc[ct][4 * g + 2] = bv.z; acc[ct][4 * g + 3] = bv.w;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&sW3[((hf * 4 + ct) * 4 + ks) * 64 + lane], m2f[ks], acc[ct], 0, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // shortcut slice landed (LDS-direct loads count in vmcnt)
      // shortcut + ReLU, in place; the packed values are also phase B's operand
      uint2 pk[4][4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2* sp = (uint2*)(stage + l31 * 256 + (((ct * 4 + g) ^ (l31 & 15)) << 4) + 8 * half);
          const uint2 xv = *sp;
          const float v0 = fmaxf(acc[ct][4 * g + 0] + bf2f(xv.x & 0xffff), 0.f), v1 = fmaxf(acc[ct][4 * g + 1] + bf2f(xv.x >> 16), 0.f);
          const float v2 = fmaxf(acc[ct][4 * g + 2] + bf2f(xv.y & 0xffff), 0.f), v3 = fmaxf(acc[ct][4 * g + 3] + bf2f(xv.y >> 16), 0.f);
          pk[ct][g] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
          *sp = pk[ct][g];
        }
      // phase B: k-step (hf, ct, j) <-> accumulator registers 8 j .. 8 j + 7 of tile ct
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 xf;
          *(uint2*)&xf = pk[ct][2 * j];
          *((uint2*)&xf + 1) = pk[ct][2 * j + 1];
          const int ksg = hf * 8 + ct * 2 + j;
#pragma unroll
          for (int rt = 0; rt < 2; ++rt)
            m1acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&sW1[(rt * 16 + ksg) * 64 + lane], xf, m1acc[rt], 0, 0, 0);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own LDS writes are visible to its reads
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + drow;
        const uint4 v = *(const uint4*)(stage + i * 1024 + lane * 16);
        if (p0 + row < a.P) *(uint4*)(a.xn + (long)(p0 + row) * COUT + hf * 128 + ((dcp ^ (row & 15)) << 3)) = v;
      }
    }
    // mid1' = relu(. + b1): [32 px][128 B] through the first 4 KiB of the stage (chunk c of row r at c ^ (r & 7))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = rt * 32 + 8 * g + 4 * half;
        const float4 bv = *(const float4*)(a.b1 + ch);
        const float v0 = fmaxf(m1acc[rt][4 * g + 0] + bv.x, 0.f), v1 = fmaxf(m1acc[rt][4 * g + 1] + bv.y, 0.f);
        const float v2 = fmaxf(m1acc[rt][4 * g + 2] + bv.z, 0.f), v3 = fmaxf(m1acc[rt][4 * g + 3] + bv.w, 0.f);
        *(uint2*)(stage + l31 * 128 + ((((ch >> 3)) ^ (l31 & 7)) << 4) + 8 * half) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + (lane >> 3), cp = lane & 7;
      const uint4 v = *(const uint4*)(stage + i * 1024 + lane * 16);
      if (p0 + row < a.P) *(uint4*)(a.m1 + (long)(p0 + row) * MID + ((cp ^ (row & 7)) << 3)) = v;
    }
  }
}

}  // namespace relnet

using namespace relnet;

// x_next = relu(conv1x1(mid2; W3, b3) + x), mid1_next = relu(conv1x1(x_next; W1n, b1n)) over P pixels (NHWC bf16, dense
// rows).  mid = 64 (res2: 64 -> 256 -> 64).  w3f = relnet_pack_w_frag of W3 [4 mid][mid]; w1f = W1n [mid][4 mid] in the
// accumulator-permuted fragment order (ops.pack_chain_w1).  Replaces two relnet_conv2d_nhwc launches
// (resnet_v1_101_rcnn_base.py: res<s><u>_branch2c + shortcut + relu, res<s><u+1>_branch2a + relu).
extern "C" int relnet_bottleneck_chain(const void* mid2, const void* x, const void* w3f, const void* w1f, const float* b3,
                                       const float* b1, void* x_next, void* mid1_next, long P, int mid, void* stream) {
  RELNET_REQUIRE(mid2 && x && w3f && w1f && b3 && b1 && x_next && mid1_next, "relnet_bottleneck_chain: null operand");
  RELNET_REQUIRE(mid == 64, "relnet_bottleneck_chain: mid = %d unsupported (64)", mid);
  RELNET_REQUIRE(P > 0 && P < (1L << 31), "relnet_bottleneck_chain: bad pixel count %ld", P);
  ChainArgs a;
  a.m2 = (const unsigned short*)mid2; a.x = (const unsigned short*)x; a.w3f = (const uint4*)w3f; a.w1f = (const uint4*)w1f;
  a.b3 = b3; a.b1 = b1; a.xn = (unsigned short*)x_next; a.m1 = (unsigned short*)mid1_next; a.P = (int)P;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)bottleneck_chain64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const long ntile = (P + 31) / 32;
  const unsigned grid = (unsigned)(ntile < 8 * 256 ? (ntile + 7) / 8 : 256);     // persistent: one workgroup per CU
  bottleneck_chain64_kernel<<<grid, 512, 65536 + 8 * 8192, (hipStream_t)stream>>>(a);
  return check_launch("relnet_bottleneck_chain");
}
