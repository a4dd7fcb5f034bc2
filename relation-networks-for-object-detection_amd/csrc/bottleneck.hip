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


// MID = 128 (res3): the weights (2 x 128 KiB) do not fit LDS, so the workgroup walks the 128-output-channel passes in
// lock step and re-loads the pass's W3 / W1' slices (32 KiB each, LDS-direct) between two barriers; every wavefront keeps
// one 32-pixel tile (mid2 fragments + the mid1' accumulators) across the passes.  The exposed weight load (~2 us per pass)
// is inside the HBM time of the tile (655 KB per 256 pixels), weights come from L2 (256 KiB per 256-pixel tile set).
template <int MID>
__global__ __launch_bounds__(512) void bottleneck_chain_stream_kernel(ChainArgs a) {
  constexpr int COUT = 4 * MID, KS = MID / 16, RT = MID / 32, NP = COUT / 128;
  constexpr int W3B = 4 * KS * 1024, W1B = RT * 8 * 1024;     // bytes per pass
  constexpr int MROW = MID * 2, MCH = MID / 8;                 // mid1' staging: bytes per pixel row, 16-byte chunks per row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const uint4* sW3 = (const uint4*)smem;                       // [4 ct][KS][64 lanes]
  const uint4* sW1 = (const uint4*)(smem + W3B);               // [RT][8 k-steps of this pass][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  unsigned char* stage = smem + W3B + W1B + wave * 8192;
  const int ntile = (a.P + 31) / 32, nset = (ntile + 7) / 8;
  const int drow = lane >> 4, dcp = lane & 15;
  for (int set = blockIdx.x; set < nset; set += gridDim.x) {
    const int p0 = (set * 8 + wave) * 32;                      // >= P for the idle waves of the last set: loads clamp, stores are masked
    const int px = min(p0 + l31, a.P - 1);
    bf16x8 m2f[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) m2f[ks] = *(const bf16x8*)(a.m2 + (long)px * MID + 16 * ks + 8 * half);
    f32x16 m1acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m1acc[rt][r] = 0.f;
#pragma unroll 1
    for (int hf = 0; hf < NP; ++hf) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave is done with the previous pass's weights / its stage
#pragma unroll
      for (int i = 0; i < (4 * KS + 7) / 8; ++i) {
        const int q = wave + 8 * i;
        if (q < 4 * KS) __builtin_amdgcn_global_load_lds((gas_ptr)(a.w3f + ((long)(hf * 4 * KS + q) * 64 + lane)), (las_ptr)(smem + q * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        const int q = wave + 8 * i, rt = q >> 3, kk = q & 7;
        __builtin_amdgcn_global_load_lds((gas_ptr)(a.w1f + ((long)(rt * (COUT / 16) + hf * 8 + kk) * 64 + lane)), (las_ptr)(smem + W3B + q * 1024), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + drow;
        const unsigned short* src = a.x + (long)min(p0 + row, a.P - 1) * COUT + hf * 128 + ((dcp ^ (row & 15)) << 3);
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(stage + i * 1024), 16, 0, 0);
      }
      f32x16 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");     // weights of this pass (all waves' parts) and the shortcut slice landed
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&sW3[(ct * KS + ks) * 64 + lane], m2f[ks], acc[ct], 0, 0, 0);
      // per 32-channel tile: shortcut + ReLU in place, then straight into the second product (the packed values of one tile
      // are the only live copy: register budget)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2* sp = (uint2*)(stage + l31 * 256 + (((ct * 4 + g) ^ (l31 & 15)) << 4) + 8 * half);
          const uint2 xv = *sp;
          const float4 bv = *(const float4*)(a.b3 + hf * 128 + ct * 32 + 8 * g + 4 * half);
          const float v0 = fmaxf(acc[ct][4 * g + 0] + bv.x + bf2f(xv.x & 0xffff), 0.f), v1 = fmaxf(acc[ct][4 * g + 1] + bv.y + bf2f(xv.x >> 16), 0.f);
          const float v2 = fmaxf(acc[ct][4 * g + 2] + bv.z + bf2f(xv.y & 0xffff), 0.f), v3 = fmaxf(acc[ct][4 * g + 3] + bv.w + bf2f(xv.y >> 16), 0.f);
          pk[g] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
          *sp = pk[g];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 xf;
          *(uint2*)&xf = pk[2 * j];
          *((uint2*)&xf + 1) = pk[2 * j + 1];
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            m1acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&sW1[(rt * 8 + ct * 2 + j) * 64 + lane], xf, m1acc[rt], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int row = 4 * i + drow;
        const uint4 v = *(const uint4*)(stage + i * 1024 + lane * 16);
        if (p0 + row < a.P) *(uint4*)(a.xn + (long)(p0 + row) * COUT + hf * 128 + ((dcp ^ (row & 15)) << 3)) = v;
      }
    }
    // mid1' = relu(. + b1): [32 px][MID] through the wave's stage (chunk c of row r at c ^ (r & (MCH - 1)))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int ch = rt * 32 + 8 * g + 4 * half;
        const float4 bv = *(const float4*)(a.b1 + ch);
        const float v0 = fmaxf(m1acc[rt][4 * g + 0] + bv.x, 0.f), v1 = fmaxf(m1acc[rt][4 * g + 1] + bv.y, 0.f);
        const float v2 = fmaxf(m1acc[rt][4 * g + 2] + bv.z, 0.f), v3 = fmaxf(m1acc[rt][4 * g + 3] + bv.w, 0.f);
        *(uint2*)(stage + l31 * MROW + (((ch >> 3) ^ (l31 & (MCH - 1))) << 4) + 8 * half) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    constexpr int RPI = 1024 / MROW;                           // pixel rows per 1 KiB store instruction
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int row = RPI * i + lane / MCH, cp = lane % MCH;
      const uint4 v = *(const uint4*)(stage + i * 1024 + lane * 16);
      if (p0 + row < a.P) *(uint4*)(a.m1 + (long)(p0 + row) * MID + ((cp ^ (row & (MCH - 1))) << 3)) = v;
    }
  }
}

}  // namespace relnet

using namespace relnet;

// x_next = relu(conv1x1(mid2; W3, b3) + x), mid1_next = relu(conv1x1(x_next; W1n, b1n)) over P pixels (NHWC bf16, dense
// rows).  mid = 64 (res2: 64 -> 256 -> 64) or 128 (res3: 128 -> 512 -> 128).  w3f = relnet_pack_w_frag of W3 [4 mid][mid]; w1f = W1n [mid][4 mid] in the
// accumulator-permuted fragment order (ops.pack_chain_w1).  Replaces two relnet_conv2d_nhwc launches
// (resnet_v1_101_rcnn_base.py: res<s><u>_branch2c + shortcut + relu, res<s><u+1>_branch2a + relu).
extern "C" int relnet_bottleneck_chain(const void* mid2, const void* x, const void* w3f, const void* w1f, const float* b3,
                                       const float* b1, void* x_next, void* mid1_next, long P, int mid, void* stream) {
  RELNET_REQUIRE(mid2 && x && w3f && w1f && b3 && b1 && x_next && mid1_next, "relnet_bottleneck_chain: null operand");
  RELNET_REQUIRE(mid == 64 || mid == 128, "relnet_bottleneck_chain: mid = %d unsupported (64, 128)", mid);
  RELNET_REQUIRE(P > 0 && P < (1L << 31), "relnet_bottleneck_chain: bad pixel count %ld", P);
  ChainArgs a;
  a.m2 = (const unsigned short*)mid2; a.x = (const unsigned short*)x; a.w3f = (const uint4*)w3f; a.w1f = (const uint4*)w1f;
  a.b3 = b3; a.b1 = b1; a.xn = (unsigned short*)x_next; a.m1 = (unsigned short*)mid1_next; a.P = (int)P;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)bottleneck_chain64_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)bottleneck_chain_stream_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const long ntile = (P + 31) / 32;
  const unsigned grid = (unsigned)(ntile < 8 * 256 ? (ntile + 7) / 8 : 256);     // persistent: one workgroup per CU
  if (mid == 64) bottleneck_chain64_kernel<<<grid, 512, 65536 + 8 * 8192, (hipStream_t)stream>>>(a);
  else bottleneck_chain_stream_kernel<128><<<grid, 512, 65536 + 8 * 8192, (hipStream_t)stream>>>(a);
  return check_launch("relnet_bottleneck_chain");
}

// ---------------------------------------------------------------------------------------
// 3x3 convolution of the 64-channel bottleneck (res2*_branch2b: stride 1, pad 1, BN folded, ReLU) with the input tile
// and its halo resident in LDS.  The implicit-GEMM kernel fills LDS once per tap (9 x the input through L2 -> LDS,
// 2.3 GB per launch at B = 54, the measured bound of its 297 us); here a (8+2) x (32+2)-pixel halo tile is fetched ONCE
// (1.33 x the input) and the nine taps are row offsets into it.
// Workgroup = 8 x 32 output pixels, persistent over tiles with the next halo tile in flight; wave (c, r) owns output
// channels 32 c .. 32 c + 31 for pixel rows 2 r, 2 r + 1.  Its 36 weight fragments (32 channels x 576) stay in VGPRs for
// the life of the workgroup, so LDS serves only the pixel operand.
// ---------------------------------------------------------------------------------------
namespace relnet {

__device__ __attribute__((aligned(16))) const unsigned int g_zero_chunk[4] = {0u, 0u, 0u, 0u};

struct Halo3Args {
  const unsigned short* in;    // [B][H][W][64] bf16, dense
  const uint4* wf;             // W [64][576] (k = tap * 64 + c) in fragment order (relnet_pack_w_frag)
  const float* bias;           // [64]
  unsigned short* out;         // [B][H][W][64]
  int B, H, W, relu;
  int tiles_x, tiles_y;        // ceil(W / 32), ceil(H / 8)
};

constexpr int kHaloW = 34, kHaloSlots = 40;                 // halo columns used / allocated per row (5 x 8 pixels)
constexpr int kHaloBytes = 10 * kHaloSlots * 128;           // 50 DMA instructions x 1 KiB

__global__ __launch_bounds__(512) void conv3x3_c64_halo_kernel(Halo3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sOut = smem + 2 * kHaloBytes;               // [256 px][128 B], chunk c of pixel p at slot c ^ ((p >> 1) & 7)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keeps tile / row arithmetic on the scalar unit
  const int half = lane >> 5, l31 = lane & 31;
  const int chh = wave & 1, rg = wave >> 1;                  // channel half, row group
  bf16x8 wfrag[36];
#pragma unroll
  for (int ks = 0; ks < 36; ++ks) wfrag[ks] = *(const bf16x8*)&a.wf[(chh * 36 + ks) * 64 + lane];
  float* sBias = (float*)(sOut + 32768);                     // [64]
  if (tid < 64) sBias[tid] = a.bias[tid];
  const int per_img = a.tiles_x * a.tiles_y;
  const int ntile = a.B * per_img;

  // halo tile in LDS: [10 rows][40 pixel slots][128 B] (34 used per row; 5 one-KiB DMA instructions per row), chunk c of
  // slot q at position c ^ ((q >> 1) & 7), q = row * 40 + column.  Everything per-instruction is wave-uniform (scalar)
  // except (lane >> 3, lane & 7): no per-lane address state survives between tiles (spilled addresses would be reloaded
  // through scratch, whose vmcnt wait would serialise the LDS-direct loads behind each other -- measured 9 us per tile).
  auto issue_halo = [&](int t, unsigned char* buf) {
    asm volatile("" : "+s"(t));                             // opaque: no loop-carried (and then spilled) per-lane pointers
    const int b = t / per_img, ty = (t % per_img) / a.tiles_x, tx = t % a.tiles_x;
    const int y0 = ty * 8 - 1, x0 = tx * 32 - 1;
    const unsigned short* img = a.in + (long)b * a.H * a.W * 64;
    int ln = lane;
    asm volatile("" : "+v"(ln));                            // opaque too: the per-lane parts are recomputed per tile, not kept live
    const int r8 = ln >> 3, slot = ln & 7;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int j = wave + 8 * i;                            // DMA instruction j: halo row j / 5, pixel slots 8 (j % 5) ..
      if (j >= 50) break;
      const int hy = j / 5, part = j - 5 * hy;
      const int y = y0 + hy, hx = part * 8 + r8, x = x0 + hx;
      const int sw = (4 * (hy + part) + (r8 >> 1)) & 7;      // ((hy * 40 + hx) >> 1) & 7
      const bool ok = hx < kHaloW && y >= 0 && y < a.H && x >= 0 && x < a.W;
      const void* src = ok ? (const void*)(img + ((long)y * a.W + x) * 64 + ((slot ^ sw) << 3)) : (const void*)g_zero_chunk;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(buf + j * 1024), 16, 0, 0);
    }
  };

  int t = blockIdx.x;
  if (t < ntile) issue_halo(t, smem);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int cur = 0;
  for (; t < ntile; t += gridDim.x, cur ^= 1) {
    // the next tile's halo is requested first and is only waited for after this tile's arithmetic; this tile's stores are
    // issued after that wait and drain during the NEXT tile's arithmetic (loads and stores share vmcnt and complete out of
    // order with respect to each other, so the only safe wait is vmcnt(0): it is placed where both have had a tile's time)
    const int tn = t + gridDim.x;
    if (tn < ntile) issue_halo(tn, smem + (cur ^ 1) * kHaloBytes);
    const unsigned char* hb = smem + cur * kHaloBytes;
    int lq = l31;
    asm volatile("" : "+v"(lq));                            // the 72 swizzled read offsets are recomputed per tile, not held in VGPRs
    f32x16 acc[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rr][r] = 0.f;
    {
      // 36 k-steps (tap, 16-channel block); the two pixel rows alternate (two independent MFMA chains) and the fragments
      // of step s + 2 are requested before the MFMAs of step s
      auto frag = [&](int s_, int rr) -> bf16x8 {
        const int tap = s_ >> 2, kb = s_ & 3, dy = tap / 3, dx = tap % 3;
        const int hp = (2 * rg + rr + dy) * kHaloSlots + lq + dx;
        return *(const bf16x8*)(hb + hp * 128 + (((2 * kb + half) ^ ((hp >> 1) & 7)) << 4));
      };
      bf16x8 pq[3][2];
      pq[0][0] = frag(0, 0); pq[0][1] = frag(0, 1);
      pq[1][0] = frag(1, 0); pq[1][1] = frag(1, 1);
#pragma unroll
      for (int s_ = 0; s_ < 36; ++s_) {
        if (s_ + 2 < 36) { pq[(s_ + 2) % 3][0] = frag(s_ + 2, 0); pq[(s_ + 2) % 3][1] = frag(s_ + 2, 1); }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[s_], pq[s_ % 3][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[s_], pq[s_ % 3][1], acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // bias -> ReLU -> bf16 -> staging [pixel][64 channels]
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      const int p = (2 * rg + rr) * 32 + l31;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bq = *(const float4*)(sBias + chh * 32 + 8 * g + 4 * half);
        float v[4] = {acc[rr][4 * g] + bq.x, acc[rr][4 * g + 1] + bq.y, acc[rr][4 * g + 2] + bq.z, acc[rr][4 * g + 3] + bq.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = a.relu ? fmaxf(v[e], 0.f) : v[e];
        *(uint2*)(sOut + p * 128 + (((chh * 4 + g) ^ ((p >> 1) & 7)) << 4) + 8 * half) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // both channel halves of every pixel are staged
    {
      const int b = t / per_img, ty = (t % per_img) / a.tiles_x, tx = t % a.tiles_x;
      unsigned short* img = a.out + (long)b * a.H * a.W * 64;
      const unsigned char* so = sOut + wave * 4096 + lane * 16;
      const uint4 v0 = *(const uint4*)so, v1 = *(const uint4*)(so + 1024), v2 = *(const uint4*)(so + 2048), v3 = *(const uint4*)(so + 3072);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // next halo landed (this wave's part); staging read
      auto put = [&](int i, const uint4& v) {
        const int p = 8 * (wave * 4 + i) + (lane >> 3), slot = lane & 7;
        const int y = ty * 8 + (p >> 5), x = tx * 32 + (p & 31);
        if (y < a.H && x < a.W) *(uint4*)(img + ((long)y * a.W + x) * 64 + ((slot ^ ((p >> 1) & 7)) << 3)) = v;
      };
      put(0, v0); put(1, v1); put(2, v2); put(3, v3);
    }
    __builtin_amdgcn_s_barrier();                           // every wave's part of the next halo is in LDS; staging is free
  }
}

}  // namespace relnet

// 3x3 / stride 1 / pad 1 convolution + bias (+ ReLU) of a dense 64-channel NHWC bf16 tensor (res2*_branch2b with BN folded;
// resnet_v1_101_rcnn_base.py:52-56).  w_frag = relnet_pack_w_frag of the packed weight [64][9 * 64] (k = (r * 3 + s) * 64 + c).
extern "C" int relnet_conv3x3_c64(const void* in, const void* w_frag, const float* bias, int relu, void* out, int B, int H,
                                  int W, void* stream) {
  RELNET_REQUIRE(in && w_frag && bias && out, "relnet_conv3x3_c64: null operand");
  RELNET_REQUIRE(B > 0 && H > 0 && W > 0 && (long)B * H * W < (1L << 31), "relnet_conv3x3_c64: bad geometry B=%d H=%d W=%d", B, H, W);
  Halo3Args a;
  a.in = (const unsigned short*)in; a.wf = (const uint4*)w_frag; a.bias = bias; a.out = (unsigned short*)out;
  a.B = B; a.H = H; a.W = W; a.relu = relu;
  a.tiles_x = (W + 31) / 32; a.tiles_y = (H + 7) / 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)conv3x3_c64_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const long ntile = (long)B * a.tiles_x * a.tiles_y;
  const unsigned grid = (unsigned)(ntile < 256 ? ntile : 256);
  conv3x3_c64_halo_kernel<<<grid, 512, 2 * kHaloBytes + 32768 + 256, (hipStream_t)stream>>>(a);
  return check_launch("relnet_conv3x3_c64");
}
