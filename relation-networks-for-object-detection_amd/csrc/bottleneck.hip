// Pixel-wise chain across a residual-block boundary of the ResNet trunk (reference graph:
// relation_rcnn/symbols/resnet_v1_101_rcnn_base.py, e.g. res2a_branch2c .. res2b_branch2a, :66-84):
//
//     x_next = relu(W3 . mid2 + b3 + x)            1x1 expand (BN folded) + identity shortcut + ReLU of block n
//     mid1'  = relu(W1' . x_next + b1')            1x1 reduce (BN folded) + ReLU of block n+1
//
// Both are 1x1 convolutions, i.e. per-pixel products, so block n+1's reduce can consume x_next while it is still in
// registers: the 256- (512-) channel activation is written once and NOT read back by a separate reduce kernel.  For the
// HBM-bound stages that is the dominant saving (res2 at B = 54: expand 535 us + reduce 256 us as two GEMM launches ->
// 461 us; DESIGN.md section 4 / 4a have the measured figures).  The same kernel without the second product is the
// expand + shortcut + ReLU layer of every residual unit (res2 .. res5), bit-identical to the implicit-GEMM launch.
//
// One wavefront = 32 pixels, in passes of 64 output channels:
//   phase A  acc^T[cout][px] = W3 (A operand, rows = cout, fragments from LDS) x mid2^T (B operand: lane = pixel, 8 channels
//            per k-step, read straight from HBM once per tile and kept in VGPRs);
//   shortcut the 32 px x 64-channel slice of x arrives in LDS by global_load_lds (1 KiB per instruction = 8 pixel rows of
//            128 B, 16-byte chunks XOR-swizzled on the SOURCE side so that the per-lane 8-byte reads spread over the banks),
//            one pass ahead; relu(acc + b3 + x) is written back IN PLACE as bf16 and leaves as 16-byte coalesced rows
//            one pass later;
//   phase B  the same packed bf16 values are the B operand of the second product (the contraction index cout is permuted
//            identically in the accumulator registers and in the pre-packed W1' fragments, as in the attention kernel),
//            mid1'^T[c][px] accumulates over the passes, then bias + ReLU + LDS transpose + coalesced rows.
// Rounding points are those of the two-launch path (x_next is rounded to bf16 before the reduce product).
#include "common.h"

namespace relnet {

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;

// LDS-direct load issued from inline assembly: the compiler tracks no LDS-DMA store for it, so it does not put its own
// s_waitcnt vmcnt(0) in front of the next ds_read (which would drain the prefetched slices at once).  Ordering by hand at the call
// sites: counted vmcnt + barrier before a read (same helper as in gemm.hip).
__device__ __forceinline__ void chain_glds16(const void* src, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory", "m0");
}

struct ChainArgs {
  const unsigned short* m2;   // [P][MID]   3x3 output of block n (bf16)
  const unsigned short* x;    // [P][4 MID] shortcut = input of block n
  const uint4* w3f;           // W3 [4 MID][MID] in fragment order (relnet_pack_w_frag)
  const uint4* w1f;           // W1' [MID][4 MID], fragment order with the accumulator permutation (ops.pack_chain_w1)
  const float* b3;            // [4 MID]
  const float* b1;            // [MID]
  const unsigned short* xin;  // PROJ: [P][MID] input of the unit (its projection shortcut is computed here, no x operand)
  const uint4* wpf;           // PROJ: Wp [4 MID][MID] (branch1) in fragment order
  unsigned short* xn;         // [P][4 MID]
  unsigned short* m1;         // [P][MID]
  int P;
  int dbg;                    // timing ablations of chain256_roles_kernel (relnet_chain_debug; results are then WRONG): 1 = half of the weight
                              // loads, 2 = no weight loads, 4 = no shortcut-slice loads, 8 = no global stores
};

// One kernel for both widths, in passes of 64 output channels of the expand product:
//   MID = 64  (res2, STREAM = false): W3 / W1' (32 KiB each) stay in LDS for the life of the workgroup and the eight
//             wavefronts run independently (no barrier after the weight load);
//   MID = 128 (res3, STREAM = true):  the weights (2 x 128 KiB) do not fit, so the workgroup walks the passes in lock step and
//             the pass's W3 / W1' slices (16 KiB each) go through a two-slot LDS ring, requested one pass ahead.
// Per wavefront the shortcut slice of pass p+1 (32 px x 64 channels, 4 KiB, second stage buffer) is requested and the
// finished slice of pass p-1 is stored BEFORE the arithmetic of pass p, so both are in flight during it; loads and stores
// share vmcnt and complete out of order with respect to each other, hence the single vmcnt(0) at the top of a pass.
// Nothing on the vector-memory path is issued inside the arithmetic (biases come from LDS): a load there would make the
// compiler wait for the prefetch in front of it.
// KSPLIT > 1 (MID = 512, res5; expand-only): the k range of a pass is cut in KSPLIT sub-steps so that a ring slot stays 32 KiB.
// (measured and dropped, r04: vector-memory roles per wavefront for this form -- four wavefronts stream weights, four move the activations
//  of all eight tiles, the mid-pass step only drains weight loads: 19.62 vs 19.59 ms per 54-image step, no gain; res5's expand is bound
//  by its 2 MB weight stream per 256 pixels through the L2 -> LDS fill path)
// MID = 256 with the second product (res4): chain256_roles_kernel below.
// PROJ (MID = 64, resident weights; the first unit of res2, whose shortcut is a 1x1 projection of the unit's 64-channel input):
//   x_next = relu(W3 . mid2 + Wp . x_in + (b3 + bp)) -- the projection is four more k-steps of the same accumulators instead of a
//   separate convolution that writes a 4 MID-channel map (1.04 GB at 54 images) for this kernel to read back; no shortcut slice,
//   one stage buffer per wave (written, then flushed in the same pass).  The shortcut is not rounded to bf16 on the way.
template <int MID, bool STREAM, bool REDUCE = true, int KSPLIT = 1, bool PROJ = false>       // REDUCE = false: only x_next (no next reduce)
__global__ __launch_bounds__(512) void bottleneck_chain_kernel(ChainArgs a) {
  static_assert(!PROJ || (!STREAM && KSPLIT == 1), "the projection form exists for the resident-weight kernel");
  static_assert(KSPLIT == 1 || (STREAM && !REDUCE), "k-split passes exist for the streamed expand-only form");
  constexpr int COUT = 4 * MID, KS = MID / 16, KSS = KS / KSPLIT, RT = MID / 32, NP = COUT / 64;
  static_assert(!REDUCE || MID <= 128, "MID = 256 with the second product is chain256_roles_kernel");
  constexpr int W3P = 2 * KSS * 1024, W1P = REDUCE ? 4 * RT * 1024 : 0;     // bytes of one (sub-)step's W3 / W1' slice
  constexpr int WPP = PROJ ? W3P : 0;                                         // ... and of its Wp slice
  constexpr int PASSB = W3P + WPP + W1P;
  constexpr int STG = PROJ ? 4096 : 8192;                                     // stage bytes per wave
  constexpr int WBYTES = STREAM ? 2 * PASSB : NP * PASSB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  unsigned char* stage = smem + WBYTES + wave * STG;           // 2 (PROJ: 1) x [32 px][128 B], chunk c of row r at c ^ ((r >> 1) & 7)
  float* sB3 = (float*)(smem + WBYTES + 8 * STG);              // [COUT]
  float* sB1 = sB3 + COUT;                                     // [MID]
  for (int i = tid; i < COUT; i += 512) sB3[i] = a.b3[i];
  if constexpr (REDUCE) for (int i = tid; i < MID; i += 512) sB1[i] = a.b1[i];
  if constexpr (!STREAM) {                                      // resident layout: pass-major, [pass][W3 slice | W1' slice]
    for (int i = tid; i < NP * PASSB / 16; i += 512) {
      const int p = i / (PASSB / 16), r = i % (PASSB / 16);
      uint4 v;
      if (r < W3P / 16) v = a.w3f[(long)p * (W3P / 16) + r];
      else if (r < (W3P + WPP) / 16) v = a.wpf[(long)p * (W3P / 16) + (r - W3P / 16)];
      else { const int q = (r - (W3P + WPP) / 16) >> 6, rt = q >> 2, kk = q & 3; v = a.w1f[((long)rt * (COUT / 16) + p * 4 + kk) * 64 + (r & 63)]; }
      ((uint4*)smem)[i] = v;
    }
  }
  __syncthreads();
  const int ntile = (a.P + 31) / 32;
  const int drow = lane >> 3, dslot = lane & 7;                // DMA / coalesced-store role: row 8 i + drow, chunk slot dslot
  // shortcut slice of (tile base pixel p0, pass p) into stage buffer p & 1
  auto issue_x = [&](int p0, int p) {
    if constexpr (PROJ) return;
    unsigned char* sb = stage + (p & 1) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + drow;
      // (32-bit byte offsets from the tensor base -- the host checks P * COUT * 2 < 2^32: half the address registers per row)
      const unsigned off = ((unsigned)min(p0 + row, a.P - 1) * COUT + p * 64 + ((dslot ^ ((row >> 1) & 7)) << 3)) * 2u;
      __builtin_amdgcn_global_load_lds((gas_ptr)((const unsigned char*)a.x + off), (las_ptr)(sb + i * 1024), 16, 0, 0);
    }
  };
  // (STREAM) this wave's share of the weights of step st = pass * KSPLIT + k-part into ring slot st & 1
  auto issue_w = [&](int st) {
    if constexpr (STREAM) {
      const int p = st / KSPLIT, kh = st % KSPLIT;
      unsigned char* wb = smem + (st & 1) * PASSB;
#pragma unroll
      for (int i = 0; i < (2 * KSS + (REDUCE ? 4 * RT : 0)) / 8; ++i) {
        const int q = wave + 8 * i;
        if (q < 2 * KSS) {
          const int ct = q / KSS, ks = q % KSS;
          __builtin_amdgcn_global_load_lds((gas_ptr)(a.w3f + ((long)((p * 2 + ct) * KS + kh * KSS + ks) * 64 + lane)), (las_ptr)(wb + q * 1024), 16, 0, 0);
        } else {
          const int q1 = q - 2 * KSS, rt = q1 >> 2, kk = q1 & 3;
          __builtin_amdgcn_global_load_lds((gas_ptr)(a.w1f + ((long)(rt * (COUT / 16) + p * 4 + kk) * 64 + lane)), (las_ptr)(wb + W3P + q1 * 1024), 16, 0, 0);
        }
      }
    }
  };
  // stores of a finished 32 px x 64 channel slice of x_next from stage buffer p & 1
  auto flush = [&](int p0, int p) {
    const unsigned char* sb = stage + (PROJ ? 0 : (p & 1) * 4096) + lane * 16;
    const uint4 v0 = *(const uint4*)sb, v1 = *(const uint4*)(sb + 1024), v2 = *(const uint4*)(sb + 2048), v3 = *(const uint4*)(sb + 3072);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    auto put = [&](int i, const uint4& v) {
      const int row = 8 * i + drow;
      if (p0 + row < a.P) *(uint4*)((unsigned char*)a.xn + ((unsigned)(p0 + row) * COUT + p * 64 + ((dslot ^ ((row >> 1) & 7)) << 3)) * 2u) = v;
    };
    put(0, v0); put(1, v1); put(2, v2); put(3, v3);
  };

  // tile sequence of this wavefront: independent tiles (resident) or lock-step sets of 8 tiles (streaming)
  const int t_first = blockIdx.x * 8 + wave;
  const int t_step = gridDim.x * 8;
  const int n_iter = STREAM ? ((ntile + 7) / 8 - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x
                            : (ntile - t_first + t_step - 1) / t_step;      // (streaming: idle waves of the last set still take part)
  if (n_iter <= 0) return;
  issue_x(t_first * 32, 0);
  issue_w(0);
  for (int it = 0; it < n_iter; ++it) {
    const int p0 = (t_first + it * t_step) * 32;               // >= P for idle waves: loads clamp, stores are masked
    const int px = min(p0 + l31, a.P - 1);
    bf16x8 m2f[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) m2f[ks] = *(const bf16x8*)(a.m2 + (long)px * MID + 16 * ks + 8 * half);
    bf16x8 xf[PROJ ? KS : 1];
    if constexpr (PROJ) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xf[ks] = *(const bf16x8*)(a.xin + (long)px * MID + 16 * ks + 8 * half);
    }
    f32x16 m1acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m1acc[rt][r] = 0.f;
#pragma unroll 1
    for (int p = 0; p < NP; ++p) {
      f32x16 acc[2];
#pragma unroll
      for (int kh = 0; kh < KSPLIT; ++kh) {
        const int st = p * KSPLIT + kh;
        if constexpr (STREAM) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");   // step st landed everywhere; everyone left step st-1
        else if constexpr (!PROJ) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (PROJ: no LDS-DMA in flight; its stores need no wait)
        if (kh == 0) {
          if (!PROJ && p > 0) flush(p0, p - 1);
          if (p + 1 < NP) issue_x(p0, p + 1);
          else if (it + 1 < n_iter) issue_x(p0 + t_step * 32, 0);
        }
        if (st + 1 < NP * KSPLIT) issue_w(st + 1);
        else if (it + 1 < n_iter) issue_w(0);
        const uint4* w3 = (const uint4*)(smem + (STREAM ? (st & 1) : st) * PASSB);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
          if (kh == 0)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KSS; ++ks)
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&w3[(ct * KSS + ks) * 64 + lane], m2f[kh * KSS + ks], acc[ct], 0, 0, 0);
          if constexpr (PROJ) {
            const uint4* wp = (const uint4*)((const unsigned char*)w3 + W3P);
#pragma unroll
            for (int ks = 0; ks < KSS; ++ks)
              acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&wp[(ct * KSS + ks) * 64 + lane], xf[ks], acc[ct], 0, 0, 0);
          }
        }
      }
      const uint4* w1 = (const uint4*)(smem + (STREAM ? (p & 1) : p) * PASSB + W3P + WPP);      // (REDUCE implies KSPLIT == 1: step == pass)
      unsigned char* sb = stage + (PROJ ? 0 : (p & 1) * 4096);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        // bias + shortcut + ReLU in place; the packed values feed the second product directly
        uint2 pk[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          uint2* sp = (uint2*)(sb + l31 * 128 + (((ct * 4 + g) ^ ((l31 >> 1) & 7)) << 4) + 8 * half);
          const uint2 xv = PROJ ? make_uint2(0u, 0u) : *sp;
          const float4 bv = *(const float4*)(sB3 + p * 64 + ct * 32 + 8 * g + 4 * half);
          const float v0 = fmaxf(acc[ct][4 * g + 0] + bv.x + bf2f(xv.x & 0xffff), 0.f), v1 = fmaxf(acc[ct][4 * g + 1] + bv.y + bf2f(xv.x >> 16), 0.f);
          const float v2 = fmaxf(acc[ct][4 * g + 2] + bv.z + bf2f(xv.y & 0xffff), 0.f), v3 = fmaxf(acc[ct][4 * g + 3] + bv.w + bf2f(xv.y >> 16), 0.f);
          pk[g] = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
          *sp = pk[g];
        }
        if constexpr (REDUCE) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            bf16x8 xf;
            *(uint2*)&xf = pk[2 * j];
            *((uint2*)&xf + 1) = pk[2 * j + 1];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt)
              m1acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)&w1[(rt * 4 + ct * 2 + j) * 64 + lane], xf, m1acc[rt], 0, 0, 0);
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slice is complete in LDS (same-wave readers only)
      __builtin_amdgcn_wave_barrier();
      if constexpr (PROJ) { flush(p0, p); __builtin_amdgcn_wave_barrier(); }      // single stage buffer: out before the next pass writes it
    }
    if constexpr (!PROJ) flush(p0, NP - 1);
    if constexpr (!REDUCE) continue;
    // mid1' = relu(. + b1), 64 channels at a time through the buffer that was just flushed (the other one is receiving
    // the next tile's first slice)
    unsigned char* sb = stage + (PROJ ? 0 : ((NP - 1) & 1) * 4096);
#pragma unroll
    for (int hc = 0; hc < RT / 2; ++hc) {
#pragma unroll
      for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int rt = 2 * hc + r2;
          const float4 bv = *(const float4*)(sB1 + rt * 32 + 8 * g + 4 * half);
          const float v0 = fmaxf(m1acc[rt][4 * g + 0] + bv.x, 0.f), v1 = fmaxf(m1acc[rt][4 * g + 1] + bv.y, 0.f);
          const float v2 = fmaxf(m1acc[rt][4 * g + 2] + bv.z, 0.f), v3 = fmaxf(m1acc[rt][4 * g + 3] + bv.w, 0.f);
          *(uint2*)(sb + l31 * 128 + (((r2 * 4 + g) ^ ((l31 >> 1) & 7)) << 4) + 8 * half) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const unsigned char* sl = sb + lane * 16;
      const uint4 v0 = *(const uint4*)sl, v1 = *(const uint4*)(sl + 1024), v2 = *(const uint4*)(sl + 2048), v3 = *(const uint4*)(sl + 3072);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      auto put = [&](int i, const uint4& v) {
        const int row = 8 * i + drow;
        if (p0 + row < a.P) *(uint4*)(a.m1 + (long)(p0 + row) * MID + hc * 64 + ((dslot ^ ((row >> 1) & 7)) << 3)) = v;
      };
      put(0, v0); put(1, v1); put(2, v2); put(3, v3);
    }
  }
}




// ---------------------------------------------------------------------------------------------------------------------------
// res4 form of the two-product chain (MID = 256: 256 -> 1024 + shortcut + ReLU, then the next unit's 1024 -> 256 + ReLU), r04,
// with ONE ROLE PER WAVEFRONT.  Every wavefront doing everything (bottleneck_chain_kernel's structure with the passes cut in two
// 32-channel steps so that the weight slots stay 32 KB: built and measured first) needs 64 (mid2 fragments) + 128 (mid1'
// accumulators) + 16 registers of essential state per wavefront of 256: the compiler spilled the address arithmetic of the activation
// loads into the step loop
// (a scratch reload -> s_waitcnt vmcnt(0) between every two HBM loads) and had one register set left for weight fragments (ds_read ->
// lgkmcnt(0) -> MFMA, 32 times per step): 21.6 ms per 54-image step against 20.0 ms with this kernel on the same box (the two
// separate launches: 22.0).  Here a workgroup walks sets of FOUR 32-pixel tiles; tile T belongs to two wavefronts:
//   A-wave T      expand product of step g (16 MFMAs: W3 rows [32 g', +32) x mid2 fragments in registers), bias + shortcut + ReLU into
//                 the tile's stage buffer, and the stores of finished slices (they drain in the background: an A-wave never waits on vmcnt
//                 except for its mid2 fragments once per tile);
//   B-wave T + 4  reduce product of step g - 1 (16 MFMAs: the 32 channels the A-wave finished one step earlier, read back from the
//                 stage buffer, x the matching W1' fragments) into 128 accumulator registers; mid1' out once per tile.
// B-waves 4, 5 also stream the weights (W3 of step g + 1 and W1' of step g into the slot not in use: vmcnt(0) per step, L2 hits);
// B-waves 6, 7 load the shortcut slices of all four tiles TWO passes ahead into a four-deep stage ring per tile and wait with a
// COUNTED vmcnt (only loads in their queue: in-order), so HBM loads have two passes and HBM stores unlimited time to complete.
// One s_barrier per step (32 channels); same passes, rounding points and results as the one-role form.
// What bounds it (r04, 54 images, 209 us per launch = 1.63 us per step): the L2 -> LDS fill path.  Every workgroup streams the full
// 1 MB of W3 + W1' per set of four tiles -- 1.0 GB per launch, + 0.27 GB of shortcut slices through the same path -- at the 5.5 - 6 TB/s
// that path sustains (tools/fill_probe.py); HBM carries 0.66 GB (3.2 TB/s).  Eight tiles per set would halve the weight stream, but
// their mid1' accumulators alone are all the registers four B-waves have.
// LDS: weights 2 x 32 KB | stage 4 tiles x 4 x 4 KB | mid1' staging 4 x 4 KB | biases 5 KB = 149 KB.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kRolesLdsBytes = 2 * 32768 + 4 * 4 * 4096 + 4 * 4096 + (1024 + 256) * 4;      // weights | stage rings | mid1' staging | biases
__global__ __launch_bounds__(512) void chain256_roles_kernel(ChainArgs a) {
  constexpr int MID = 256, COUT = 4 * MID, KS = MID / 16, RT = MID / 32, NP = COUT / 64, NSTEP = 2 * NP, NTL = 4, NBUF = 4;
  constexpr int W3P = KS * 1024, W1P = 2 * RT * 1024, SLOT = W3P + W1P, WBYTES = 2 * SLOT;
  constexpr int STG = NBUF * 4096, M1S = WBYTES + NTL * STG, BIAS = M1S + NTL * 4096;
  static_assert(BIAS + (COUT + MID) * 4 == kRolesLdsBytes, "host launch and kernel disagree on the LDS layout");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31, drow = lane >> 3, dslot = lane & 7;
  const unsigned lds0 = (unsigned)(unsigned long)(las_ptr)smem;
  float* sB3 = (float*)(smem + BIAS);
  float* sB1 = sB3 + COUT;
  for (int i = tid; i < COUT; i += 512) sB3[i] = a.b3[i];
  for (int i = tid; i < MID; i += 512) sB1[i] = a.b1[i];
  const int ntile = (a.P + 31) / 32, nset = (ntile + NTL - 1) / NTL;
  const int n_iter = (nset - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // (idle tiles of the last set still take part)
  if (n_iter <= 0) return;
  const int G = n_iter * NSTEP, NPASS = n_iter * NP;            // steps / passes of this workgroup
  auto tile_p0 = [&](int it, int T) { return (((int)blockIdx.x + it * (int)gridDim.x) * NTL + T) * 32; };     // >= P for idle tiles: loads clamp, stores are masked
  // shortcut slice of global pass Pg for tile T -> stage buffer Pg % NBUF
  auto issue_x = [&](int T, int Pg) {
    if (a.dbg & 4) return;
    const int p0 = tile_p0(Pg / NP, T), p = Pg % NP;
    unsigned char* sb = smem + WBYTES + T * STG + (Pg % NBUF) * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = 8 * i + drow;
      const unsigned off = ((unsigned)min(p0 + row, a.P - 1) * COUT + p * 64 + ((dslot ^ ((row >> 1) & 7)) << 3)) * 2u;
      chain_glds16((const unsigned char*)a.x + off, __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(sb - smem) + i * 1024));
    }
  };
  // weights of the slot step g + 1 reads: W3 rows of step g + 1 (fragments 0..15) and W1' k-steps of step g (fragments 16..31); loader l of 2
  auto issue_w = [&](int g, int l) {
    if (a.dbg & 2) return;
    unsigned char* wb = smem + ((g + 1) & 1) * SLOT;
    const int s3 = (g + 1) % NSTEP, s1 = g % NSTEP;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if ((a.dbg & 1) && (i & 1)) continue;
      const int q = l + 2 * i;
      if (q < KS) {
        if (g + 1 < G)
          chain_glds16((const unsigned char*)a.w3f + ((unsigned)((s3 * KS + q) * 1024) + lane * 16u), __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wb - smem) + q * 1024));
      } else if (g >= 0 && g < G) {
        const int q1 = q - KS, rt = q1 >> 1, j = q1 & 1;
        chain_glds16((const unsigned char*)a.w1f + ((unsigned)((rt * (COUT / 16) + (s1 >> 1) * 4 + (s1 & 1) * 2 + j) * 1024) + lane * 16u), __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(wb - smem) + W3P + q1 * 1024));
      }
    }
  };
  // prologue: W3 of step 0 (both weight loaders: their share of slot 0), shortcut slices of passes 0 and 1
  if (wave == 4 || wave == 5) issue_w(-1, wave - 4);
  if (wave >= 6) {
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) { issue_x(2 * (wave - 6) + tt, 0); }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) { issue_x(2 * (wave - 6) + tt, 1); }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");       // biases in LDS (this also drains the prologue loads once)

  if (wave < 4) {
    // ================================================= A-wave: tile T = wave =================================================
    const int T = wave;
    unsigned char* const stg = smem + WBYTES + T * STG;
    bf16x8 m2f[KS];
#pragma unroll 1
    for (int g = 0; g <= G + 1; ++g) {
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // (this wave's stage writes of the previous step are in LDS)
      if (g < G) {
        const int st = g % NSTEP, p = st >> 1, ct = st & 1, Pg = g >> 1;
        if (st == 0) {
          const int px = min(tile_p0(g / NSTEP, T) + l31, a.P - 1);
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) m2f[ks] = *(const bf16x8*)(a.m2 + (long)px * MID + 16 * ks + 8 * half);
          __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0) HERE, once per tile, as a real instruction the compiler's wait-count pass sees: left to it (or
                                                   // hidden in an asm statement) counted vmcnt waits land on the main path of every step and drain the stores
        }
        const uint4* w3 = (const uint4*)(smem + (g & 1) * SLOT);
        // all sixteen weight fragments of the step up front (64 registers this role has to spare): the expand product is ONE dependent
        // MFMA chain (measured: the step time does not depend on it -- see the bound below -- but the MFMA pipe is released earlier)
        bf16x8 wf[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = *(const bf16x8*)&w3[ks * 64 + lane];
        unsigned char* sb = stg + (Pg % NBUF) * 4096;
        uint2 xv[4];
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) xv[gq] = *(const uint2*)(sb + l31 * 128 + (((ct * 4 + gq) ^ ((l31 >> 1) & 7)) << 4) + 8 * half);
        __builtin_amdgcn_sched_barrier(0);                   // (the scheduler otherwise sinks the reads back between the MFMAs)
        f32x16 ac;
#pragma unroll
        for (int r = 0; r < 16; ++r) ac[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
          ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks], m2f[ks], ac, 0, 0, 0);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          uint2* sp = (uint2*)(sb + l31 * 128 + (((ct * 4 + gq) ^ ((l31 >> 1) & 7)) << 4) + 8 * half);
          const float4 bv = *(const float4*)(sB3 + p * 64 + ct * 32 + 8 * gq + 4 * half);
          const float v0 = fmaxf(ac[4 * gq + 0] + bv.x + bf2f(xv[gq].x & 0xffff), 0.f), v1 = fmaxf(ac[4 * gq + 1] + bv.y + bf2f(xv[gq].x >> 16), 0.f);
          const float v2 = fmaxf(ac[4 * gq + 2] + bv.z + bf2f(xv[gq].y & 0xffff), 0.f), v3 = fmaxf(ac[4 * gq + 3] + bv.w + bf2f(xv[gq].y >> 16), 0.f);
          *sp = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
        }
      }
      if ((g & 1) && g >= 3) {
        // slice Pf is complete (A: step 2 Pf + 1) and consumed (B: step 2 Pf + 2): out, 16-byte coalesced rows
        const int Pf = (g - 3) >> 1, p0 = tile_p0(Pf / NP, T), p = Pf % NP;
        const unsigned char* sb = stg + (Pf % NBUF) * 4096 + lane * 16;
        const uint4 v0 = *(const uint4*)sb, v1 = *(const uint4*)(sb + 1024), v2 = *(const uint4*)(sb + 2048), v3 = *(const uint4*)(sb + 3072);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        auto put = [&](int i, const uint4& v) {
          const int row = 8 * i + drow;
          if (p0 + row < a.P && !(a.dbg & 8)) *(uint4*)((unsigned char*)a.xn + ((unsigned)(p0 + row) * COUT + p * 64 + ((dslot ^ ((row >> 1) & 7)) << 3)) * 2u) = v;
        };
        put(0, v0); put(1, v1); put(2, v2); put(3, v3);
      }
    }
  } else {
    // ================================================= B-wave: tile T = wave - 4 =============================================
    const int T = wave - 4;
    const unsigned char* const stg = smem + WBYTES + T * STG;
    unsigned char* const m1s = smem + M1S + T * 4096;
    f32x16 m1acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int r = 0; r < 16; ++r) m1acc[rt][r] = 0.f;
    bool drain = false;                      // this wave has stores in its queue (mid1' rows): the next counted wait must be a full one
#pragma unroll 1
    for (int g = 0; g <= G + 1; ++g) {
      if (wave < 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // weights of step g (and of B's step g - 1) landed
      else if (!(g & 1) && (g >> 1) < NPASS) {                                           // slice of pass g / 2 landed; the next one may be in flight
        if (!drain && (g >> 1) + 1 < NPASS) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        drain = false;
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (wave < 6) issue_w(g, wave - 4);
      else if (!(g & 1) && (g >> 1) + 2 < NPASS) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) issue_x(2 * (wave - 6) + tt, (g >> 1) + 2);
      }
      if (g >= 1 && g <= G) {
        const int h = g - 1, st = h % NSTEP, ct = st & 1, Pg = h >> 1;                 // the step the A-wave finished before this barrier
        const uint4* w1 = (const uint4*)(smem + (g & 1) * SLOT + W3P);
        const unsigned char* sb = stg + (Pg % NBUF) * 4096;
        bf16x8 xf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          *(uint2*)&xf[j] = *(const uint2*)(sb + l31 * 128 + (((ct * 4 + 2 * j) ^ ((l31 >> 1) & 7)) << 4) + 8 * half);
          *((uint2*)&xf[j] + 1) = *(const uint2*)(sb + l31 * 128 + (((ct * 4 + 2 * j + 1) ^ ((l31 >> 1) & 7)) << 4) + 8 * half);
        }
        bf16x8 wf[2][RT];                                      // (all sixteen fragments up front, as in the A-wave)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) wf[j][rt] = *(const bf16x8*)&w1[(rt * 2 + j) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
            m1acc[rt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[j][rt], xf[j], m1acc[rt], 0, 0, 0);
        if (st == NSTEP - 1) {
          // mid1' = relu(. + b1) of the tile that just ended, 64 channels at a time through this wave's staging buffer
          const int p0 = tile_p0(h / NSTEP, T);
#pragma unroll
          for (int hc = 0; hc < RT / 2; ++hc) {
#pragma unroll
            for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
              for (int gq = 0; gq < 4; ++gq) {
                const int rt = 2 * hc + r2;
                const float4 bv = *(const float4*)(sB1 + rt * 32 + 8 * gq + 4 * half);
                const float v0 = fmaxf(m1acc[rt][4 * gq + 0] + bv.x, 0.f), v1 = fmaxf(m1acc[rt][4 * gq + 1] + bv.y, 0.f);
                const float v2 = fmaxf(m1acc[rt][4 * gq + 2] + bv.z, 0.f), v3 = fmaxf(m1acc[rt][4 * gq + 3] + bv.w, 0.f);
                *(uint2*)(m1s + l31 * 128 + (((r2 * 4 + gq) ^ ((l31 >> 1) & 7)) << 4) + 8 * half) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
              }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            const unsigned char* sl = m1s + lane * 16;
            const uint4 v0 = *(const uint4*)sl, v1 = *(const uint4*)(sl + 1024), v2 = *(const uint4*)(sl + 2048), v3 = *(const uint4*)(sl + 3072);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
            auto put = [&](int i, const uint4& v) {
              const int row = 8 * i + drow;
              if (p0 + row < a.P && !(a.dbg & 8)) *(uint4*)(a.m1 + (long)(p0 + row) * MID + hc * 64 + ((dslot ^ ((row >> 1) & 7)) << 3)) = v;
            };
            put(0, v0); put(1, v1); put(2, v2); put(3, v3);
          }
#pragma unroll
          for (int rt = 0; rt < RT; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) m1acc[rt][r] = 0.f;
          drain = true;
        }
      }
    }
  }
}

}  // namespace relnet

using namespace relnet;

// x_next = relu(conv1x1(mid2; W3, b3) + x), mid1_next = relu(conv1x1(x_next; W1n, b1n)) over P pixels (NHWC bf16, dense
// rows).  mid = 64 (res2: 64 -> 256 -> 64) or 128 (res3: 128 -> 512 -> 128).  w3f = relnet_pack_w_frag of W3 [4 mid][mid]; w1f = W1n [mid][4 mid] in the
// accumulator-permuted fragment order (ops.pack_chain_w1).  Replaces two relnet_conv2d_nhwc launches
// (resnet_v1_101_rcnn_base.py: res<s><u>_branch2c + shortcut + relu, res<s><u+1>_branch2a + relu).
static int g_chain_dbg = 0;        // timing ablations of chain256_roles_kernel (ChainArgs::dbg): results are wrong while it is non-zero
extern "C" void relnet_chain_debug(int flags) { g_chain_dbg = flags; }

extern "C" int relnet_bottleneck_chain(const void* mid2, const void* x, const void* w3f, const void* w1f, const float* b3,
                                       const float* b1, void* x_next, void* mid1_next, long P, int mid, void* stream) {
  RELNET_REQUIRE(mid2 && x && w3f && b3 && x_next, "relnet_bottleneck_chain: null operand");
  RELNET_REQUIRE((w1f && b1 && mid1_next) || (!w1f && !b1 && !mid1_next), "relnet_bottleneck_chain: w1f, b1 and mid1_next are given together (or all NULL: expand + shortcut + ReLU only)");
  RELNET_REQUIRE(mid == 64 || mid == 128 || mid == 256 || (mid == 512 && !mid1_next), "relnet_bottleneck_chain: mid = %d unsupported (64, 128, 256; 512 without the reduce product)", mid);
  RELNET_REQUIRE(P > 0 && P * 8 * mid < (1L << 32), "relnet_bottleneck_chain: bad pixel count %ld (the 4 mid-channel map must stay below 4 GiB)", P);
  ChainArgs a;
  a.m2 = (const unsigned short*)mid2; a.x = (const unsigned short*)x; a.w3f = (const uint4*)w3f; a.w1f = (const uint4*)w1f;
  a.b3 = b3; a.b1 = b1; a.xn = (unsigned short*)x_next; a.m1 = (unsigned short*)mid1_next; a.P = (int)P;
  a.xin = nullptr; a.wpf = nullptr; a.dbg = g_chain_dbg;
  static relnet::PerDeviceOnce attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<64, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<128, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<256, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<512, true, false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)chain256_roles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const long ntile = (P + 31) / 32;
  const unsigned grid = (unsigned)(ntile < 8 * 256 ? (ntile + 7) / 8 : 256);     // persistent: one workgroup per CU
  const size_t lds = 65536 /* weights: resident (mid 64) or two ring slots (mid 128) */ + 65536 /* 8 x 2 stage buffers */ + (size_t)5 * mid * 4;
  if (mid1_next) {
    if (mid == 64) bottleneck_chain_kernel<64, false><<<grid, 512, lds, (hipStream_t)stream>>>(a);
    else if (mid == 128) bottleneck_chain_kernel<128, true><<<grid, 512, lds, (hipStream_t)stream>>>(a);
    else {
      const long nset = (ntile + 3) / 4;
      chain256_roles_kernel<<<(unsigned)(nset < 256 ? nset : 256), 512, relnet::kRolesLdsBytes, (hipStream_t)stream>>>(a);
    }
  } else {
    if (mid == 64) bottleneck_chain_kernel<64, false, false><<<grid, 512, lds, (hipStream_t)stream>>>(a);
    else if (mid == 128) bottleneck_chain_kernel<128, true, false><<<grid, 512, lds, (hipStream_t)stream>>>(a);
    else if (mid == 256) bottleneck_chain_kernel<256, true, false><<<grid, 512, lds, (hipStream_t)stream>>>(a);
    else bottleneck_chain_kernel<512, true, false, 2><<<grid, 512, lds, (hipStream_t)stream>>>(a);
  }
  return check_launch("relnet_bottleneck_chain");
}

// First unit of res2 (projection shortcut, stride 1): x_next = relu(conv1x1(mid2; W3) + conv1x1(x_in; Wp) + b3p), b3p = b3 + bp, and
// (optionally) mid1_next = relu(conv1x1(x_next; W1n, b1n)) in one kernel: the branch1 convolution of the reference graph
// (resnet_v1_101_rcnn_base.py: res2a_branch1 + bn2a_branch1) becomes four more k-steps of the expand product.  mid = 64 only;
// w3f / wpf = relnet_pack_w_frag of W3 / Wp [256][64]; x_in [P][64] dense bf16.
extern "C" int relnet_bottleneck_chain_proj(const void* mid2, const void* x_in, const void* w3f, const void* wpf, const void* w1f,
                                            const float* b3p, const float* b1, void* x_next, void* mid1_next, long P, int mid,
                                            void* stream) {
  RELNET_REQUIRE(mid2 && x_in && w3f && wpf && b3p && x_next, "relnet_bottleneck_chain_proj: null operand");
  RELNET_REQUIRE((w1f && b1 && mid1_next) || (!w1f && !b1 && !mid1_next), "relnet_bottleneck_chain_proj: w1f, b1 and mid1_next are given together (or all NULL)");
  RELNET_REQUIRE(mid == 64, "relnet_bottleneck_chain_proj: mid = %d unsupported (64)", mid);
  RELNET_REQUIRE(P > 0 && P * 8 * mid < (1L << 32), "relnet_bottleneck_chain_proj: bad pixel count %ld (x_next is addressed with 32-bit byte offsets: the 4 mid-channel map must stay below 4 GiB)", P);
  ChainArgs a;
  a.m2 = (const unsigned short*)mid2; a.x = nullptr; a.w3f = (const uint4*)w3f; a.w1f = (const uint4*)w1f;
  a.b3 = b3p; a.b1 = b1; a.xn = (unsigned short*)x_next; a.m1 = (unsigned short*)mid1_next; a.P = (int)P;
  a.xin = (const unsigned short*)x_in; a.wpf = (const uint4*)wpf; a.dbg = 0;
  static relnet::PerDeviceOnce attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<64, false, true, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)bottleneck_chain_kernel<64, false, false, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const long ntile = (P + 31) / 32;
  const unsigned grid = (unsigned)(ntile < 8 * 256 ? (ntile + 7) / 8 : 256);
  const size_t lds = (size_t)4 * (8192 + 8192 + (mid1_next ? 8192 : 0)) + 8 * 4096 + (size_t)5 * mid * 4;
  if (mid1_next) bottleneck_chain_kernel<64, false, true, 1, true><<<grid, 512, lds, (hipStream_t)stream>>>(a);
  else bottleneck_chain_kernel<64, false, false, 1, true><<<grid, 512, lds, (hipStream_t)stream>>>(a);
  return check_launch("relnet_bottleneck_chain_proj");
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
  static relnet::PerDeviceOnce attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)conv3x3_c64_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const long ntile = (long)B * a.tiles_x * a.tiles_y;
  const unsigned grid = (unsigned)(ntile < 256 ? ntile : 256);
  conv3x3_c64_halo_kernel<<<grid, 512, 2 * kHaloBytes + 32768 + 256, (hipStream_t)stream>>>(a);
  return check_launch("relnet_conv3x3_c64");
}

// (Measured and dropped: the same halo-resident form for the 256-channel res4 3x3 layers -- halo tile per 64- or 128-channel
//  part, weights [256][64] through a 2-, 3- or 4-slot LDS ring -- reaches 178-186 us against 166 us of the implicit-GEMM ring
//  kernel in isolation.  With 8 fragment MFMAs per 6 LDS fragment reads and a workgroup barrier per weight slab the LDS-read +
//  MFMA rate, not the L2 -> LDS fill it removes, is what bounds those layers.)
