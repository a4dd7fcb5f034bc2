// C = A W^T (+bias) (+residual) (ReLU) -- the FullyConnected contractions of the hot path
// (fc_new_1/fc_new_2/cls_score/bbox_pred: SYM_REL:254-280; query_i/key_i/linear_out_i:
// SYM_REL:120-129,146-150 of the reference; MXNet FullyConnected = x W^T + b, W [out,in]).
//
// Both operands are K-contiguous ("NT"), which is exactly the MFMA A/B fragment order, so
// no transposes are staged anywhere.
//   * bf16 kernel: 128x128x64 (or 64x64x64) workgroup tile, 4 waves (2x2), LDS double
//     buffer with a 16-byte-chunk XOR swizzle, v_mfma_f32_32x32x16_bf16, fp32 accumulate.
//   * f32 kernel (parity path): v_mfma_f32_32x32x2_f32 fed straight from global/L2 -- the
//     contraction index is permuted so that each lane's 8 consecutive floats of a row are
//     its 8 k-slots (A and W use the same permutation, so the product is unchanged).
// Batched-strided over blockIdx.z (used for the per-image V*Wout^T product).
#include "common.h"
#include <type_traits>

namespace relnet {

struct GemmArgs {
  const void* A; long lda; long strideA;
  const void* W; long ldw; long strideW;
  void* C; long ldc; long strideC;
  const float* bias;       // nullptr or fp32 vector
  const void* resid;       // nullptr or same layout/dtype as C
  int M, N, K;
  int bias_mode;           // 0 none, 1 per output column (N), 2 per output row (M)
  int relu;
  // implicit-GEMM convolution (CONV kernels only): A is an NHWC image batch, row m of the
  // GEMM is output pixel (b, oy, ox), k = (r*S + s)*Cin + ic; W is [Cout][R][S][Cin].
  int cH, cW, cCin, cHout, cWout, cR, cS, cStride, cDil, cPad;
  long cPix, cImg;         // element strides between pixels / images of the input
  int n_loop;              // column tiles walked by one workgroup (row-panel mode), >= 1
  int xcd_swizzle;         // 1: remap the linear workgroup id so that every XCD owns a contiguous run of tiles
  const void* Wf;          // optional: W pre-packed in MFMA fragment order (relnet_pack_w_frag), used by gemm_panelw_kernel
  long long* phase_ts;     // debug (relnet_gemm_debug_phase_ts): per workgroup 8 words -- wall clock (100 MHz) at entry, k-loop start, k-loop end,
                           // exit, and the shader cycles entry -> k-loop end; nullptr = off
  int korder;              // ring kernels, R*S > 1: 1 = walk k as (channel chunk, tap) instead of (tap, channel chunk) -- see launch_ring
  const void* mask;        // relu == 3 (relnet_gemm_nt_mask, MASK instantiations of gemm_nt_bf16_kernel only): C = (A W^T + resid) where mask > 0, else 0;
                           // same layout / dtype as C (bf16).  The backward of `x_next = relu(conv(..) + x)`: the shortcut gradient rides as
                           // `resid`, the saved forward activation as `mask` -- one launch instead of GEMM + relnet_relu_bwd
  // split-K (gemm_nt_bf16_kernel, batch 1, n_loop 1): blockIdx.z = which of `ksplit` contiguous runs of k-slabs this workgroup multiplies.
  // Every workgroup leaves its fp32 partial tile in kpart [ksplit][M][N]; the LAST one to arrive at a tile (kcnt, one counter per tile,
  // left at zero again) sums the partials in split order -- the result does not depend on the arrival order -- and runs the epilogue.
  int ksplit;
  float* kpart;
  unsigned int* kcnt;
};

template <typename TOUT> __device__ __forceinline__ void store_out(TOUT* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }
template <typename TOUT> __device__ __forceinline__ float load_out(const TOUT* p);
template <> __device__ __forceinline__ float load_out<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_out<unsigned short>(const unsigned short* p) { return bf2f(*p); }

// The residual operand of an epilogue: relu == 2 turns it into a ReLU MASK (backward of a fused conv + ReLU: the data gradient
// is zeroed where the saved forward activation is not positive) instead of a summand.
__device__ __forceinline__ float res_apply(float v, float r, int relu) { return relu == 2 ? (r > 0.f ? v : 0.f) : v + r; }

template <typename TOUT>
__device__ __forceinline__ void epilogue_tile(const GemmArgs& g, TOUT* C, const TOUT* R,
                                              const f32x16& acc, int row0, int col0, int lane) {
  const int col = col0 + (lane & 31);
  if (col >= g.N) return;
  const float bcol = (g.bias_mode == 1) ? g.bias[col] : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = row0 + mfma32_row(r, lane);
    if (row < g.M) {
      float v = acc[r] + bcol;
      if (g.bias_mode == 2) v += g.bias[row];
      if (R) v = res_apply(v, load_out<TOUT>(R + (long)row * g.ldc + col), g.relu);
      if (g.relu == 1) v = fmaxf(v, 0.f);
      store_out<TOUT>(C + (long)row * g.ldc + col, v);
    }
  }
}

// Epilogue for accumulators computed with SWAPPED MFMA operands (D = W_frag x A_frag): lane
// owns output row m = row0 + (lane & 31) and, per register group gq, four CONSECUTIVE output
// columns col0 + 8 gq + 4 (lane >> 5) + 0..3 -> 8-byte (bf16) / 16-byte (f32) stores and
// residual loads instead of one 2-byte element per lane.
template <typename TOUT>
__device__ __forceinline__ void epilogue_tile_t(const GemmArgs& g, TOUT* C, const TOUT* R,
                                                const f32x16& acc, int row0, int col0, int lane) {
  const int m = row0 + (lane & 31);
  if (m >= g.M) return;
  const float brow = (g.bias_mode == 2) ? g.bias[m] : 0.f;
  const bool vec_ok = (g.ldc & 3) == 0;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const int n = col0 + 8 * gq + 4 * (lane >> 5);
    if (n >= g.N) continue;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = acc[4 * gq + e] + brow;
    TOUT* cp = C + (long)m * g.ldc + n;
    const TOUT* rp = R ? R + (long)m * g.ldc + n : nullptr;
    if (vec_ok && n + 3 < g.N) {
      if (g.bias_mode == 1) {
        const float4 bv = *(const float4*)(g.bias + n);
        v[0] += bv.x; v[1] += bv.y; v[2] += bv.z; v[3] += bv.w;
      }
      if constexpr (sizeof(TOUT) == 2) {
        if (rp) {
          const uint2 rv = *(const uint2*)rp;
          v[0] = res_apply(v[0], bf2f(rv.x & 0xffff), g.relu); v[1] = res_apply(v[1], bf2f(rv.x >> 16), g.relu); v[2] = res_apply(v[2], bf2f(rv.y & 0xffff), g.relu); v[3] = res_apply(v[3], bf2f(rv.y >> 16), g.relu);
        }
        if (g.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *(uint2*)cp = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      } else {
        if (rp) {
          const float4 rv = *(const float4*)rp;
          v[0] = res_apply(v[0], rv.x, g.relu); v[1] = res_apply(v[1], rv.y, g.relu); v[2] = res_apply(v[2], rv.z, g.relu); v[3] = res_apply(v[3], rv.w, g.relu);
        }
        if (g.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (n + e < g.N) {
          float x = v[e];
          if (g.bias_mode == 1) x += g.bias[n + e];
          if (rp) x = res_apply(x, load_out<TOUT>(rp + e), g.relu);
          if (g.relu == 1) x = fmaxf(x, 0.f);
          store_out<TOUT>(cp + e, x);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// bf16 in, fp32 accumulate.  BM x BN workgroup tile, TM x TN MFMA tiles per wave.
// ---------------------------------------------------------------------------------------
// 16 zero bytes: source of out-of-range rows / padded convolution taps for the LDS-direct loads
__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

typedef const __attribute__((address_space(1))) void* gas_ptr;
typedef __attribute__((address_space(3))) void* las_ptr;
// LDS-direct load issued from inline assembly: the compiler then tracks no LDS-DMA store, so it does not put its own
// s_waitcnt vmcnt(0) in front of the first ds_read it cannot disambiguate from the DMA destination (which would drain a counted
// multi-slab pipeline once per slab).  Ordering is entirely by hand at the call sites: counted vmcnt + barrier before a read.
__device__ __forceinline__ void glds16_asm(const void* src, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(lds_byte_addr) : "memory", "m0");
}

// BM x BN workgroup tile, WM x WN waves (each wave: TM x TN MFMA tiles of 32x32), BK = 64,
// two LDS stages filled by global_load_lds.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// F16: the 16-bit operands are IEEE half instead of bfloat16 (v_mfma_f32_32x32x16_f16 -- same 8-pass instruction, same staging, same
// LDS image; fp32 outputs only).  Exists for BASELINE configs[4]'s "fp16 MFMA stress" wording: relnet_gemm_nt_f16 measures that an
// fp16 layer runs at the rate of its bf16 twin on gfx950, which is why the FPN configuration is timed with bf16 operands.
template <int BM, int BN, int WM, int WN, typename TOUT, int MODE, int KU = 1, bool MASK = false, bool F16 = false>
__global__ __launch_bounds__(64 * WM * WN) void gemm_nt_bf16_kernel(GemmArgs g) {
  constexpr bool CONV = MODE == 1;       // NHWC implicit GEMM, Cin % 64 == 0
  constexpr bool STEM = MODE == 2;       // 7x7/2 stem on a zero-padded NHWC4 image (see relnet_stem_conv7)
  constexpr int BK = 64;
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int A_GROUPS = BM / (8 * NW), B_GROUPS = BN / (8 * NW);   // 8-row (1 KiB) groups per wave
  static_assert(TM >= 1 && TN >= 1 && A_GROUPS >= 1 && B_GROUPS >= 1, "tile too small for the wave grid");
  constexpr int STAGE = (BM + BN) * BK * 2;            // bytes of one k-slab; a pipeline stage holds KU of them
  constexpr int CLD = BN + 4;                          // padded fp32 row of the epilogue band
  constexpr int BAND = 32 * WM;                        // rows written per epilogue pass
  constexpr int LDS_BYTES = (2 * KU * STAGE > BAND * CLD * 4) ? 2 * KU * STAGE : BAND * CLD * 4;
  // ONE shared array (a second __shared__ object makes hipcc drain vmcnt before every ds_read)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  // Workgroup b runs on XCD b % 8 (observed dispatch rule).  The column tiles of one row panel share the A rows, so they
  // should sit on ONE XCD (one L2) close together in time: give every XCD a contiguous run of the (row panel, column
  // group) sequence (bijective for any grid size).  Measured before: the 4 column tiles of the res4 expand convolutions
  // re-fetched A from the fabric 4x (profiles/conv_pmc.json: 820 MB moved for 596 MB of operands).
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_swizzle) {
    const int gx = gridDim.x, nwg = gx * gridDim.y;
    const int id = bx + gx * by, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    by = t / gx; bx = t - by * gx;
  }
  const int m0 = by * BM;
  const bool splitk = g.ksplit > 1;
  const int zb = splitk ? 0 : blockIdx.z, ks = splitk ? (int)blockIdx.z : 0;
  const unsigned short* A = (const unsigned short*)g.A + (long)zb * g.strideA;
  const unsigned short* W = (const unsigned short*)g.W + (long)zb * g.strideW;
  TOUT* C = (TOUT*)g.C + (long)zb * g.strideC;
  const TOUT* R = g.resid ? (const TOUT*)g.resid + (long)zb * g.strideC : nullptr;

  // LDS-direct staging (global_load_lds, 16 B per lane): one instruction fills 8 rows x 128 B,
  // lane l -> LDS (row l>>3, slot l&7).  The bank-conflict swizzle slot = chunk ^ (row & 7) is
  // applied on the SOURCE side: lane l fetches global chunk (l&7) ^ (l>>3) of its row.
  const int lrow = lane >> 3, lchunk = (lane & 7) ^ (lane >> 3);
  const unsigned short* arow[A_GROUPS];
  const unsigned short* brow[B_GROUPS];
  int ciy[A_GROUPS], cix[A_GROUPS];
#pragma unroll
  for (int j = 0; j < A_GROUPS; ++j) {
    const int gr = m0 + (wave * A_GROUPS + j) * 8 + lrow;
    if constexpr (CONV) {
      const int hw = g.cHout * g.cWout;
      const int b = gr / hw, rem = gr - b * hw;
      const int oy = rem / g.cWout, ox = rem - oy * g.cWout;
      arow[j] = A + (long)b * g.cImg + lchunk * 8;
      ciy[j] = (gr < g.M) ? oy * g.cStride - g.cPad : -(1 << 28);     // row >= M: never in bounds
      cix[j] = ox * g.cStride - g.cPad;
    } else if constexpr (STEM) {
      // padded image [B][Hp][Wp][4]: output (oy, ox) reads rows 2oy..2oy+7, 32 contiguous elements
      // (8 pixels x 4 ch) from pixel 2ox; k-tile kt covers tap rows 2kt, 2kt+1; chunk c of the tile
      // = tap row 2kt + (c >> 2), elements 8 (c & 3)...
      const int hw = g.cHout * g.cWout;
      const int gr2 = gr < g.M ? gr : g.M - 1;
      const int b = gr2 / hw, rem = gr2 - b * hw;
      const int oy = rem / g.cWout, ox = rem - oy * g.cWout;
      arow[j] = A + (long)b * g.cImg + ((long)(2 * oy + (lchunk >> 2)) * g.cW + 2 * ox) * 4 + (lchunk & 3) * 8;
      ciy[j] = 0; cix[j] = 0;
    } else {
      arow[j] = (gr < g.M) ? A + (long)gr * g.lda + lchunk * 8 : nullptr;
    }
  }
  // Row-panel mode (n_loop > 1): this workgroup walks n_loop consecutive column tiles of its row
  // tile, so its output rows are written as long contiguous runs and the A rows are re-read
  // from its own XCD's L2 instead of by workgroups scattered over the chip.
  for (int nt = 0; nt < g.n_loop; ++nt) {
  const int n0 = (bx * g.n_loop + nt) * BN;
  if (n0 >= g.N) break;
  if (nt > 0) __syncthreads();                         // previous tile's epilogue band fully read
#pragma unroll
  for (int j = 0; j < B_GROUPS; ++j) {
    const int gr = n0 + (wave * B_GROUPS + j) * 8 + lrow;
    brow[j] = (gr < g.N) ? W + (long)gr * g.ldw + lchunk * 8 : nullptr;
  }
  auto stage = [&](int kt, int slot) {                  // k-slab kt into slab slot `slot` (= stage * KU + position in the stage)
    const int buf = slot;
    const int k0 = kt * BK;
    int tr = 0, ts = 0, ic0 = k0;
    if constexpr (CONV) {
      const int tap = k0 / g.cCin;
      ic0 = k0 - tap * g.cCin;
      tr = tap / g.cS; ts = tap - tr * g.cS;
    }
    unsigned char* la = lds + buf * STAGE + wave * (A_GROUPS * 1024);
    unsigned char* lb = lds + buf * STAGE + BM * BK * 2 + wave * (B_GROUPS * 1024);
#pragma unroll
    for (int j = 0; j < A_GROUPS; ++j) {
      const void* src;
      if constexpr (CONV) {
        const int iy = ciy[j] + tr * g.cDil, ix = cix[j] + ts * g.cDil;
        const bool ok = (iy >= 0) && (iy < g.cH) && (ix >= 0) && (ix < g.cW);
        src = ok ? (const void*)(arow[j] + ((long)iy * g.cW + ix) * g.cPix + ic0) : (const void*)g_zero16;
      } else if constexpr (STEM) {
        src = (const void*)(arow[j] + (long)(2 * kt) * g.cW * 4);
      } else {
        src = arow[j] ? (const void*)(arow[j] + k0) : (const void*)g_zero16;
      }
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(la + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_GROUPS; ++j) {
      const void* src = brow[j] ? (const void*)(brow[j] + k0) : (const void*)g_zero16;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(lb + j * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // epilogue thread mapping (needed up front to prefetch the residual rows)
  constexpr int VEC = 16 / sizeof(TOUT);               // output elements per 16-byte store
  constexpr int TPR = BN / VEC;                        // threads per band row
  constexpr int RPP = NT / TPR;                        // rows per copy pass
  constexpr int NP = (BAND + RPP - 1) / RPP;           // copy passes per band
  const int tcol = (tid % TPR) * VEC, trow = tid / TPR;
  const int n = n0 + tcol;
  const bool vec_ok = ((g.ldc % VEC) == 0) && ((((size_t)C) & 15) == 0) && (!R || (((size_t)R) & 15) == 0);
  auto out_row = [&](int i, int p) {                   // global row of (band i, pass p) or -1
    const int brow_i = p * RPP + trow;
    if (brow_i >= BAND) return -1;
    const int m = m0 + (brow_i >> 5) * (BM / WM) + i * 32 + (brow_i & 31);
    return (m < g.M && n < g.N) ? m : -1;
  };
  // The residual tile is fetched NOW (16 B per thread per pass) so that its latency hides behind
  // the main loop; fetched inside the epilogue it serialises one HBM round trip per copy pass,
  // which made the residual-add 1x1 convolutions run at 1.8-2.5 TB/s.
  uint4 rpre[TM][NP];
  if (R && vec_ok && !splitk) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int m = out_row(i, p);
        rpre[i][p] = (m >= 0 && n + VEC <= g.N) ? *(const uint4*)(R + (long)m * g.ldc + n) : make_uint4(0, 0, 0, 0);
      }
  }

  // KU k-slabs per barrier: with less than one workgroup per CU (the 1-image step) a k-step costs one L2 round trip whatever it
  // carries, so two slabs per step halve the k-loop (res4 3x3 at 2 394 pixels: 36 round trips -> 18)
  int kbeg = 0, nk = g.K / BK;
  if (splitk) {                                      // this workgroup's run of k-slabs: the first (nk % ksplit) runs are one slab longer
    const int base = nk / g.ksplit, rem = nk - base * g.ksplit;
    kbeg = ks * base + (ks < rem ? ks : rem);
    nk = kbeg + base + (ks < rem ? 1 : 0);
  }
#pragma unroll
  for (int h = 0; h < KU; ++h)
    if (kbeg + h < nk) stage(kbeg + h, h);
  __syncthreads();                                   // (drains the LDS-direct loads: vmcnt(0))
  for (int kt0 = kbeg; kt0 < nk; kt0 += KU) {
    const int sb = ((kt0 - kbeg) / KU) & 1;
#pragma unroll
    for (int h = 0; h < KU; ++h)
      if (kt0 + KU + h < nk) stage(kt0 + KU + h, (sb ^ 1) * KU + h);      // async: lands while this stage is multiplied
#pragma unroll
    for (int h = 0; h < KU; ++h) {
    if (KU > 1 && kt0 + h >= nk) break;
    const unsigned char* la = lds + (sb * KU + h) * STAGE;
    const unsigned char* lb = la + BM * BK * 2;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[TM], bfr[TN];
      const int ch = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * (BM / WM) + i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(la + row * 128 + ((ch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * (BN / WN) + j * 32 + (lane & 31);
        bfr[j] = *(const bf16x8*)(lb + row * 128 + ((ch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          if constexpr (F16) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, bfr[j]), __builtin_bit_cast(f16x8, af[i]), acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);   // D = W x A: lane <-> output row
    }
    }
    __syncthreads();
  }

  // ---- epilogue: per band of 32 rows per wave-row: accumulators -> LDS (fp32, padded rows) ->
  // coalesced 16-byte stores with bias / residual / ReLU applied on the way out -------------
  float* ct = (float*)lds;
  float bv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) bv[e] = (g.bias_mode == 1 && n + e < g.N) ? g.bias[n + e] : 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i > 0) __syncthreads();                        // previous band fully copied out
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int row = wr * 32 + (lane & 31);
        const int col = wc * (BN / WN) + j * 32 + 8 * gq + 4 * (lane >> 5);
        *(float4*)(ct + row * CLD + col) = make_float4(acc[i][j][4 * gq], acc[i][j][4 * gq + 1],
                                                       acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]);
      }
    // MASK: the band's mask rows are requested before the barrier, so their latency overlaps the accumulator hand-over through LDS
    uint4 mpre[MASK ? NP : 1];
    if constexpr (MASK) {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int m = out_row(i, p);
        mpre[p] = (m >= 0 && n + VEC <= g.N) ? *(const uint4*)((const TOUT*)g.mask + (long)m * g.ldc + n) : make_uint4(0, 0, 0, 0);
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = out_row(i, p);
      if (m < 0) continue;
      const int brow_i = p * RPP + trow;
      float v[VEC];
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const float4 x = *(const float4*)(ct + brow_i * CLD + tcol + 4 * q);
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      }
      if (splitk) {                                    // fp32 partial tile (N % VEC == 0 is a launch condition)
        // device-coherent (sc1, write-through) stores: the eight L2s of the chip are not coherent with each other, and a release FENCE
        // here would write back and invalidate the whole L2 of this XCD (measured: +20 us per launch and split)
        unsigned long long* pp = (unsigned long long*)(g.kpart + ((long)ks * g.M + m) * g.N + n);
#pragma unroll
        for (int q = 0; q < VEC / 2; ++q)
          __hip_atomic_store(pp + q, ((unsigned long long)__float_as_uint(v[2 * q + 1]) << 32) | __float_as_uint(v[2 * q]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        continue;
      }
      const float brow = (g.bias_mode == 2) ? g.bias[m] : 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] += bv[e] + brow;
      TOUT* cp = C + (long)m * g.ldc + n;
      if (vec_ok && n + VEC <= g.N) {
        if constexpr (sizeof(TOUT) == 2) {
          if (R) {
            const unsigned int rw[4] = {rpre[i][p].x, rpre[i][p].y, rpre[i][p].z, rpre[i][p].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = res_apply(v[2 * e], bf2f(rw[e] & 0xffff), g.relu); v[2 * e + 1] = res_apply(v[2 * e + 1], bf2f(rw[e] >> 16), g.relu); }
          }
          if (g.relu == 1) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if constexpr (MASK) {
            const unsigned int mw[4] = {mpre[p].x, mpre[p].y, mpre[p].z, mpre[p].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              v[2 * e] = bf2f(mw[e] & 0xffff) > 0.f ? v[2 * e] : 0.f;
              v[2 * e + 1] = bf2f(mw[e] >> 16) > 0.f ? v[2 * e + 1] : 0.f;
            }
          }
          *(uint4*)cp = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
          if (R) {
            v[0] = res_apply(v[0], __uint_as_float(rpre[i][p].x), g.relu); v[1] = res_apply(v[1], __uint_as_float(rpre[i][p].y), g.relu);
            v[2] = res_apply(v[2], __uint_as_float(rpre[i][p].z), g.relu); v[3] = res_apply(v[3], __uint_as_float(rpre[i][p].w), g.relu);
          }
          if (g.relu == 1) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        const TOUT* rp = R ? R + (long)m * g.ldc + n : nullptr;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (n + e < g.N) {
            float x = v[e];
            if (rp) x = res_apply(x, load_out<TOUT>(rp + e), g.relu);
            if (g.relu == 1) x = fmaxf(x, 0.f);
            store_out<TOUT>(cp + e, x);
          }
        }
      }
    }
  }
  if (splitk) {
    // ---- split-K fix-up: the partial tile is complete in memory (sc1 stores + vmcnt(0)), count in, and let the last arriver finish the tile ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned int* flag = (unsigned int*)lds;           // (the epilogue band is dead after the barrier)
    if (tid == 0) {
      unsigned int* cnt = g.kcnt + (by * gridDim.x + bx);
      const unsigned int old = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const bool last = old == (unsigned)(g.ksplit - 1);
      if (last) __hip_atomic_store(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (nobody else touches this counter any more)
      *flag = last ? 1u : 0u;
    }
    __syncthreads();
    if (*flag == 0u) return;
    constexpr int ITEMS = BM * BN / 4;                 // float4 items of the tile
    const long plane = (long)g.M * g.N;
    for (int it = tid; it < ITEMS; it += NT) {
      const int r = it / (BN / 4), cq = it - r * (BN / 4);
      const int m = m0 + r, nn = n0 + 4 * cq;
      if (m >= g.M || nn >= g.N) continue;
      const unsigned long long* pp = (const unsigned long long*)(g.kpart + (long)m * g.N + nn);
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      // device-coherent loads (they bypass this XCD's L2), ALL requested before the first is used (one memory round trip per item instead of
      // one per split), summed in split order
      unsigned long long lo[8], hi[8];
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) {
        lo[sp] = hi[sp] = 0ull;                        // (+0.0f)
        if (sp < g.ksplit) {
          lo[sp] = __hip_atomic_load(pp + sp * (plane / 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          hi[sp] = __hip_atomic_load(pp + sp * (plane / 2) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
#pragma unroll
      for (int sp = 0; sp < 8; ++sp) {
        v[0] += __uint_as_float((unsigned)lo[sp]); v[1] += __uint_as_float((unsigned)(lo[sp] >> 32));
        v[2] += __uint_as_float((unsigned)hi[sp]); v[3] += __uint_as_float((unsigned)(hi[sp] >> 32));
      }
      const float brow = (g.bias_mode == 2) ? g.bias[m] : 0.f;
      TOUT* cp = C + (long)m * g.ldc + nn;
      const TOUT* rp = R ? R + (long)m * g.ldc + nn : nullptr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float x = v[e] + brow + (g.bias_mode == 1 ? g.bias[nn + e] : 0.f);
        if (rp) x = res_apply(x, load_out<TOUT>(rp + e), g.relu);
        if (g.relu == 1) x = fmaxf(x, 0.f);
        if constexpr (MASK) x = load_out<TOUT>((const TOUT*)g.mask + (long)m * g.ldc + nn + e) > 0.f ? x : 0.f;
        v[e] = x;
      }
      if constexpr (sizeof(TOUT) == 2) *(uint2*)cp = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
      else *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  }   // nt (row-panel loop)
}

// ---------------------------------------------------------------------------------------
// Deep-pipeline variant of the kernel above (same tile geometry, staging instruction, MFMA operand order and
// epilogue): NSTAGE LDS buffers of BK-deep k-slabs filled by global_load_lds, NSTAGE-1 slabs in flight.  One RAW
// s_barrier per slab behind a COUNTED s_waitcnt vmcnt (never 0 in steady state), so the loads of the next slabs
// stay in flight across the barrier -- __syncthreads() would drain them (its fence waits vmcnt(0) for a pending
// LDS-DMA), which is what bounds the 2-stage kernel: one slab's MFMA time (~0.85 us at 256x256x64) is shorter than
// a loaded-chip L2/HBM round trip, and with one 128 KB workgroup per CU nothing else hides it.
//   ordering (guide, "8-phase template"): slab t is read only after (own counted vmcnt) + (a barrier every wave has
//   passed); buffer (t-1) % NSTAGE is re-filled only after that same barrier, i.e. after every wave's MFMAs of slab
//   t-1 were issued, which implies its ds_reads had returned.
// LDS image: rows of BK bf16 (128 B / 64 B), 16-byte chunk c of row r stored at slot c ^ swz(r) -- applied on the
// SOURCE address of the LDS-direct load and on the fragment read.  swz is chosen so that the four 16-lane groups of
// a ds_read_b128 ({0-3,12-15,20-27} ...) hit 16 distinct 16-byte bank slots: (r >> 1) & 7 for 128-byte rows
// ((r & 7), used by the 2-stage kernel above, is 2-way conflicted: rows 0 / 24 and 12 / 20 collide), (r >> 2) & 3
// for 64-byte rows.
// ---------------------------------------------------------------------------------------
template <int BK> __device__ __forceinline__ int ring_swz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// ---------------------------------------------------------------------------------------
// SCHED = 5: hand-scheduled k-loop of the 256 x 256 x 64 ring tile (2 x 4 waves, 128 x 64 outputs per wave).
// Why: the compiler's loop for SCHED = 0 (ISA read in round 4) (a) keeps ONE register set for the activation fragments and
// waits lgkmcnt(0) right behind every ds_read -- each pair of MFMAs sits behind a full LDS round trip -- and (b) puts the
// whole address arithmetic of the next slab's LDS-direct loads (13 VALU + two exec-mask branches per load) between the
// barrier and the first MFMA of a slab, where neither wave of a SIMD has matrix work in flight.
// Here one asm statement is one k-slab:
//   * accumulators live in a[0:127] for the life of the k-loop (named literally; the compiler keeps no value there), the
//     fragments in two register sets v[72:95] / v[96:119]: the reads of k-step kk+2 are issued into the set that k-step kk
//     has just consumed, between the MFMAs of k-step kk+1, with counted lgkmcnt waits;
//   * the next slab's eight LDS-direct loads are spread over the MFMAs of k-steps 0 and 1: a lane's source is its
//     precomputed pixel pointer + ONE wave-uniform tap / chunk offset (SGPR pair), a padded tap or a row past M takes the
//     16 zero bytes through a per-row tap bit mask (5 VALU per activation load, none per filter load);
//   * one s_waitcnt vmcnt(0) + s_barrier per slab, as in SCHED = 0 (the ring has two buffers).
// Hazards handled by hand (no compiler padding inside asm): an instruction separates every m0 write from its LDS-DMA,
// a fragment register set is re-targeted by ds_read only after the last MFMA reading it has ISSUED (operands are read at
// issue, LDS data returns >= 64 cycles later), 24 wait states separate the last MFMA from the accumulator read-out.
#define RA_0 "a[0:15]"
#define RA_1 "a[16:31]"
#define RA_2 "a[32:47]"
#define RA_3 "a[48:63]"
#define RA_4 "a[64:79]"
#define RA_5 "a[80:95]"
#define RA_6 "a[96:111]"
#define RA_7 "a[112:127]"
// fragment set S (0 / 1): activation rows i = 0..3, filter rows j = 0..1
#define RFA_0_0 "v[72:75]"
#define RFA_0_1 "v[76:79]"
#define RFA_0_2 "v[80:83]"
#define RFA_0_3 "v[84:87]"
#define RFB_0_0 "v[88:91]"
#define RFB_0_1 "v[92:95]"
#define RFA_1_0 "v[96:99]"
#define RFA_1_1 "v[100:103]"
#define RFA_1_2 "v[104:107]"
#define RFA_1_3 "v[108:111]"
#define RFB_1_0 "v[112:115]"
#define RFB_1_1 "v[116:119]"
#define RMF(S, I, J, ACC) "v_mfma_f32_32x32x16_bf16 " ACC ", " RFB_##S##_##J ", " RFA_##S##_##I ", " ACC "\n\t"
// reads of one fragment set from the LDS address registers AREG (activations) / BREG (filters)
#define RRD_A(S, AREG) "ds_read_b128 " RFA_##S##_0 ", " AREG "\n\tds_read_b128 " RFA_##S##_1 ", " AREG " offset:4096\n\t" \
                       "ds_read_b128 " RFA_##S##_2 ", " AREG " offset:8192\n\tds_read_b128 " RFA_##S##_3 ", " AREG " offset:12288\n\t"
#define RRD_B(S, BREG) "ds_read_b128 " RFB_##S##_0 ", " BREG " offset:32768\n\tds_read_b128 " RFB_##S##_1 ", " BREG " offset:36864\n\t"
// next-slab activation load j through the address pair P (v[P:P+1]): mask test + pointer, select, destination, issue
#define RLA_CALC(P, P1, J) "v_and_b32 v" #P ", %[tb], %[mk" #J "]\n\tv_cmp_ne_u32 vcc, 0, v" #P "\n\tv_lshl_add_u64 v[" #P ":" #P1 "], %[ab" #J "], 0, %[so]\n\t"
#define RLA_SEL(P, P1, OFF) "v_cndmask_b32 v" #P ", %[zlo], v" #P ", vcc\n\tv_cndmask_b32 v" #P1 ", %[zhi], v" #P1 ", vcc\n\ts_add_u32 m0, %[ma], " #OFF "\n\t"
#define RLA_GO(P, P1) "global_load_lds_dwordx4 v[" #P ":" #P1 "], off\n\t"
#define RLW_M0(OFF) "s_add_u32 m0, %[mb], " #OFF "\n\t"
#define RLW_GO(J) "global_load_lds_dwordx4 %[wo" #J "], %[wb]\n\t"

#define RSLAB_HEAD                                                                        \
  "s_waitcnt vmcnt(0)\n\ts_barrier\n\t"                                                   \
  RRD_A(0, "%[la0]") RRD_B(0, "%[lb0]") RRD_A(1, "%[la1]") RRD_B(1, "%[lb1]")             \
  "v_xor_b32 v120, 0x40, %[la0]\n\tv_xor_b32 v121, 0x40, %[la1]\n\t"                      \
  "v_xor_b32 v122, 0x40, %[lb0]\n\tv_xor_b32 v123, 0x40, %[lb1]\n\t"
// k-step on set S with one filler string behind each of its eight MFMAs
#define RKSTEP(S, F0, F1, F2, F3, F4, F5, F6, F7)                                          \
  RMF(S, 0, 0, RA_0) F0 RMF(S, 0, 1, RA_1) F1 RMF(S, 1, 0, RA_2) F2 RMF(S, 1, 1, RA_3) F3 \
  RMF(S, 2, 0, RA_4) F4 RMF(S, 2, 1, RA_5) F5 RMF(S, 3, 0, RA_6) F6 RMF(S, 3, 1, RA_7) F7
#define RSLAB_LOAD                                                                         \
  RSLAB_HEAD RLA_CALC(124, 125, 0) "s_waitcnt lgkmcnt(6)\n\t"                              \
  RKSTEP(0, RLA_SEL(124, 125, 0), RLA_GO(124, 125) RLA_CALC(126, 127, 1), RLA_SEL(126, 127, 1024), RLA_GO(126, 127) RLA_CALC(124, 125, 2), \
            RLA_SEL(124, 125, 2048), RLA_GO(124, 125) RLA_CALC(126, 127, 3), RLA_SEL(126, 127, 3072), RLA_GO(126, 127))                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(1, "ds_read_b128 " RFA_0_0 ", v120\n\tds_read_b128 " RFA_0_1 ", v120 offset:4096\n\t",                                             \
            "ds_read_b128 " RFA_0_2 ", v120 offset:8192\n\tds_read_b128 " RFA_0_3 ", v120 offset:12288\n\t",                                \
            RRD_B(0, "v122"), RLW_M0(0), RLW_GO(0) RLW_M0(1024), RLW_GO(1) RLW_M0(2048), RLW_GO(2) RLW_M0(3072), RLW_GO(3))                 \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(0, "ds_read_b128 " RFA_1_0 ", v121\n\tds_read_b128 " RFA_1_1 ", v121 offset:4096\n\t",                                             \
            "ds_read_b128 " RFA_1_2 ", v121 offset:8192\n\tds_read_b128 " RFA_1_3 ", v121 offset:12288\n\t",                                \
            RRD_B(1, "v123"), "", "", "", "", "")                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(1, "", "", "", "", "", "", "", "")
#define RSLAB_LAST                                                                         \
  RSLAB_HEAD "s_waitcnt lgkmcnt(6)\n\t"                                                    \
  RKSTEP(0, "", "", "", "", "", "", "", "")                                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(1, "ds_read_b128 " RFA_0_0 ", v120\n\tds_read_b128 " RFA_0_1 ", v120 offset:4096\n\t",                                             \
            "ds_read_b128 " RFA_0_2 ", v120 offset:8192\n\tds_read_b128 " RFA_0_3 ", v120 offset:12288\n\t",                                \
            RRD_B(0, "v122"), "", "", "", "", "")                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(0, "ds_read_b128 " RFA_1_0 ", v121\n\tds_read_b128 " RFA_1_1 ", v121 offset:4096\n\t",                                             \
            "ds_read_b128 " RFA_1_2 ", v121 offset:8192\n\tds_read_b128 " RFA_1_3 ", v121 offset:12288\n\t",                                \
            RRD_B(1, "v123"), "", "", "", "", "")                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(1, "", "", "", "", "", "", "", "")                                                \
  "s_nop 15\n\ts_nop 7\n\t"
// SCHED = 6: the same hand-scheduled slab with an ASYMMETRIC ring for layers whose activation rows come from HBM (1x1 "reduce"
// convolutions, K = 1024): activations through THREE 32 KB slots, fetched TWO slabs ahead, filters (L2-resident, short latency)
// through two slots one slab ahead -- 3 x 32 + 2 x 32 KB = the whole 160 KB LDS.  Loads retire in order, so a slab issues its
// filter loads first and the (younger) activation loads of slab kt + 2 last: the wait at the head of the next slab is
// vmcnt(4) -- everything but those four activation loads -- and the HBM latency of a slab gets two slab times instead of one.
#define RKSTEP_X(...) RKSTEP(__VA_ARGS__)          // (expands argument macros that carry commas first)
#define RRD_B6(S, BREG) "ds_read_b128 " RFB_##S##_0 ", " BREG "\n\tds_read_b128 " RFB_##S##_1 ", " BREG " offset:4096\n\t"
#define RSLAB6_HEAD(VM)                                                                   \
  "s_waitcnt vmcnt(" #VM ")\n\ts_barrier\n\t"                                             \
  RRD_A(0, "%[la0]") RRD_B6(0, "%[lb0]") RRD_A(1, "%[la1]") RRD_B6(1, "%[lb1]")           \
  "v_xor_b32 v120, 0x40, %[la0]\n\tv_xor_b32 v121, 0x40, %[la1]\n\t"                      \
  "v_xor_b32 v122, 0x40, %[lb0]\n\tv_xor_b32 v123, 0x40, %[lb1]\n\t"                      \
  "s_waitcnt lgkmcnt(6)\n\t"
#define RSLAB6_K1_READS                                                                    \
  "ds_read_b128 " RFA_0_0 ", v120\n\tds_read_b128 " RFA_0_1 ", v120 offset:4096\n\t",      \
  "ds_read_b128 " RFA_0_2 ", v120 offset:8192\n\tds_read_b128 " RFA_0_3 ", v120 offset:12288\n\t", \
  RRD_B6(0, "v122")
#define RSLAB6_TAIL(NOPS)                                                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(0, "ds_read_b128 " RFA_1_0 ", v121\n\tds_read_b128 " RFA_1_1 ", v121 offset:4096\n\t",                                             \
            "ds_read_b128 " RFA_1_2 ", v121 offset:8192\n\tds_read_b128 " RFA_1_3 ", v121 offset:12288\n\t",                                \
            RRD_B6(1, "v123"), "", "", "", "", "")                                         \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP(1, "", "", "", "", "", "", "", "") NOPS
#define RSLAB6_FULL                                                                        \
  RSLAB6_HEAD(4)                                                                           \
  RKSTEP(0, RLW_M0(0), RLW_GO(0) RLW_M0(1024), RLW_GO(1) RLW_M0(2048), RLW_GO(2) RLW_M0(3072), RLW_GO(3) RLA_CALC(124, 125, 0),            \
            RLA_SEL(124, 125, 0), RLA_GO(124, 125) RLA_CALC(126, 127, 1), RLA_SEL(126, 127, 1024))                                          \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP_X(1, RSLAB6_K1_READS, RLA_GO(126, 127) RLA_CALC(124, 125, 2), RLA_SEL(124, 125, 2048), RLA_GO(124, 125) RLA_CALC(126, 127, 3),      \
            RLA_SEL(126, 127, 3072), RLA_GO(126, 127))                                     \
  RSLAB6_TAIL("")
#define RSLAB6_WONLY                                                                       \
  RSLAB6_HEAD(4)                                                                           \
  RKSTEP(0, RLW_M0(0), RLW_GO(0) RLW_M0(1024), RLW_GO(1) RLW_M0(2048), RLW_GO(2) RLW_M0(3072), RLW_GO(3), "", "", "")                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP_X(1, RSLAB6_K1_READS, "", "", "", "", "")                                           \
  RSLAB6_TAIL("")
#define RSLAB6_LAST                                                                        \
  RSLAB6_HEAD(0)                                                                           \
  RKSTEP(0, "", "", "", "", "", "", "", "")                                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                               \
  RKSTEP_X(1, RSLAB6_K1_READS, "", "", "", "", "")                                           \
  RSLAB6_TAIL("s_nop 15\n\ts_nop 7\n\t")
// (measured and dropped, r04: the same loop over a per-chunk activation WINDOW shared by the nine taps of a 3x3 layer -- tile 20, -42 % of
//  the L2 -> LDS fill bytes, border taps zeroed by per-row masks -- parity-green and SLOWER: res4 3x3 152 vs 143 us, rpn 3x3 952 vs 955 us.
//  With the hand-scheduled loop these layers run at ~1.07 PFLOP/s on random operands, 55 % MFMA-busy per slab; the fill bytes are not
//  what is left)
#define RCL_AGPR "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", RCL10(a, 1), RCL10(a, 2), RCL10(a, 3), RCL10(a, 4), RCL10(a, 5), \
                 RCL10(a, 6), RCL10(a, 7), RCL10(a, 8), RCL10(a, 9), RCL10(a, 10), RCL10(a, 11), "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127"
#define RCL10(P, B) #P #B "0", #P #B "1", #P #B "2", #P #B "3", #P #B "4", #P #B "5", #P #B "6", #P #B "7", #P #B "8", #P #B "9"
#define RCL_VTMP "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", RCL10(v, 8), RCL10(v, 9), RCL10(v, 10), RCL10(v, 11), \
                 "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127"

// a wave-uniform 64-bit value the compiler may have parked in VGPRs -> provably scalar (the "s" constraint of the slab statements)
__device__ __forceinline__ unsigned long uniform64(unsigned long v) {
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return ((unsigned long)hi << 32) | lo;
}
template <int N> __device__ __forceinline__ float agpr_read() {
  float x;
  asm volatile("v_accvgpr_read_b32 %0, a[%1]" : "=v"(x) : "n"(N));
  return x;
}
template <int B> __device__ __forceinline__ void agpr_read16(f32x16& t) {
  t[0] = agpr_read<B + 0>(); t[1] = agpr_read<B + 1>(); t[2] = agpr_read<B + 2>(); t[3] = agpr_read<B + 3>();
  t[4] = agpr_read<B + 4>(); t[5] = agpr_read<B + 5>(); t[6] = agpr_read<B + 6>(); t[7] = agpr_read<B + 7>();
  t[8] = agpr_read<B + 8>(); t[9] = agpr_read<B + 9>(); t[10] = agpr_read<B + 10>(); t[11] = agpr_read<B + 11>();
  t[12] = agpr_read<B + 12>(); t[13] = agpr_read<B + 13>(); t[14] = agpr_read<B + 14>(); t[15] = agpr_read<B + 15>();
}
#define RZ8(B) "v_accvgpr_write_b32 a" #B "0, 0\n\tv_accvgpr_write_b32 a" #B "1, 0\n\tv_accvgpr_write_b32 a" #B "2, 0\n\tv_accvgpr_write_b32 a" #B "3, 0\n\t" \
               "v_accvgpr_write_b32 a" #B "4, 0\n\tv_accvgpr_write_b32 a" #B "5, 0\n\tv_accvgpr_write_b32 a" #B "6, 0\n\tv_accvgpr_write_b32 a" #B "7, 0\n\t"
#define RZ10(B) RZ8(B) "v_accvgpr_write_b32 a" #B "8, 0\n\tv_accvgpr_write_b32 a" #B "9, 0\n\t"
__device__ __forceinline__ void agpr_zero128() {
  asm volatile("v_accvgpr_write_b32 a0, 0\n\tv_accvgpr_write_b32 a1, 0\n\tv_accvgpr_write_b32 a2, 0\n\tv_accvgpr_write_b32 a3, 0\n\tv_accvgpr_write_b32 a4, 0\n\t"
               "v_accvgpr_write_b32 a5, 0\n\tv_accvgpr_write_b32 a6, 0\n\tv_accvgpr_write_b32 a7, 0\n\tv_accvgpr_write_b32 a8, 0\n\tv_accvgpr_write_b32 a9, 0\n\t"
               RZ10(1) RZ10(2) RZ10(3) RZ10(4) RZ10(5) RZ10(6) RZ10(7) RZ10(8) RZ10(9) RZ10(10) RZ10(11) RZ8(12)
               ::: RCL_AGPR);
}

template <int N> __device__ __forceinline__ void wait_vm_barrier() {
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// RESID = false instantiations carry no residual-prefetch registers (64 VGPRs at 256x256): the MFMA-bound 3x3
// convolutions, which have no shortcut operand, keep all A / W fragments of a k-step live instead.
// SCHED = 1: fragment-pipelined schedule.  The A / W fragments of k-step kk+1 are read from LDS while the MFMAs of
// k-step kk run (two named fragment sets, order pinned with sched_barrier), and the slab hand-over (counted vmcnt +
// lgkmcnt(0) + barrier, refill of the buffer just consumed, first fragments of the next slab) sits in front of the
// LAST MFMA group of a slab instead of between slabs, so neither the LDS read latency nor the barrier skew is exposed.
// A buffer is refilled only after every wave has waited for its own reads of it (lgkmcnt(0) before the barrier).
// ABLATE (measurement only, results are garbage): 1 = no LDS fragment reads / MFMAs (fill path alone),
// 2 = no global->LDS fills after the first slab (LDS read + MFMA path alone)
// EPI = 1: register epilogue.  With the swapped MFMA operands a lane owns one output ROW and, per register group gq, four
// consecutive columns (8 gq + 4 (lane >> 5) + 0..3); v_permlane32_swap of the fp32 values of groups (2p, 2p+1) gives every
// lane EIGHT consecutive columns (16 p + 8 (lane >> 5) + 0..7), so bias / shortcut / ReLU apply on 16 contiguous bytes and
// the tile leaves as 16-byte stores without the LDS transposition band (no LDS traffic, no epilogue barriers, and the ring
// buffers stay free for the next tile's prefetch).
template <int BM, int BN, int WM, int WN, typename TOUT, int MODE, int BK, int NSTAGE, bool RESID, int SCHED = 0, int ABLATE = 0, int EPI = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_ring_kernel(GemmArgs g) {
  constexpr bool CONV = MODE == 1;
  constexpr bool STEM = MODE == 2;
  static_assert(!STEM || BK == 64, "stem mode is laid out for 64-deep slabs");
  static_assert(NSTAGE >= 2 && NSTAGE <= 4, "2..4 LDS buffers");
  constexpr int NW = WM * WN, NT = 64 * NW;
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
  constexpr int ROWB = BK * 2;                         // bytes per LDS row
  constexpr int CPR = BK / 8;                          // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;                        // rows filled by one global_load_lds (1 KiB)
  constexpr int A_GROUPS = BM / (RPI * NW), B_GROUPS = BN / (RPI * NW);
  static_assert(TM >= 1 && TN >= 1 && A_GROUPS >= 1 && B_GROUPS >= 1, "tile too small for the wave grid");
  constexpr int LPS = A_GROUPS + B_GROUPS;             // LDS-direct loads per thread per slab
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int CLD = BN + 4;
  constexpr int BAND = 32 * WM;
  constexpr int LDS_BYTES = (SCHED == 4 || SCHED == 6) ? 160 * 1024 : (EPI == 1 || NSTAGE * STAGE > BAND * CLD * 4) ? NSTAGE * STAGE : BAND * CLD * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

  const long long ts_in = g.phase_ts ? wall_clock64() : 0, cy_in = g.phase_ts ? (long long)__builtin_readcyclecounter() : 0;
  long long ts_loop = ts_in;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  int bx = blockIdx.x, by = blockIdx.y;
  if (g.xcd_swizzle) {
    const int gx = gridDim.x, nwg = gx * gridDim.y;
    const int id = bx + gx * by, xcd = id & 7, q = nwg >> 3, r = nwg & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (id >> 3);
    by = t / gx; bx = t - by * gx;
  }
  const int m0 = by * BM;
  const unsigned short* A = (const unsigned short*)g.A + (long)blockIdx.z * g.strideA;
  const unsigned short* W = (const unsigned short*)g.W + (long)blockIdx.z * g.strideW;
  TOUT* C = (TOUT*)g.C + (long)blockIdx.z * g.strideC;
  const TOUT* R = (RESID && g.resid) ? (const TOUT*)g.resid + (long)blockIdx.z * g.strideC : nullptr;

  const int lrow = lane / CPR, lslot = lane % CPR;
  const unsigned short* arow[A_GROUPS];
  const unsigned short* brow[B_GROUPS];
  int ciy[A_GROUPS], cix[A_GROUPS];
  // first tile row of staging instruction j of this wave.  SCHED = 2 stages HALF tiles: A half h = rows {wr' * 128 + h * 64 +
  // [0, 64)} (the rows of quadrant row h of both wave rows), W half h = rows {wc' * 64 + h * 32 + [0, 32)}; instruction
  // j = 2 h + ii covers rows (wave * 2 + ii) * 8 .. + 7 of the 128-row half.
  auto a_base_row = [&](int j) {
    if constexpr (SCHED == 2) { const int idx = (wave * 2 + (j & 1)) * RPI; return (idx >> 6) * 128 + (j >> 1) * 64 + (idx & 63); }
    else return (wave * A_GROUPS + j) * RPI;
  };
  auto b_base_row = [&](int j) {
    if constexpr (SCHED == 2) { const int idx = (wave * 2 + (j & 1)) * RPI; return (idx >> 5) * 64 + (j >> 1) * 32 + (idx & 31); }
    else return (wave * B_GROUPS + j) * RPI;
  };
  // (image, oy, ox) of a tile row: one pair of integer divisions for this lane's first row, the other staging instructions of the
  // wave are RPI rows further each -- stepped with carries (the divisions of all A_GROUPS rows and the tap-mask loops below were
  // 4.3 - 4.9 us of every workgroup's 70 us at 54 images: relnet_gemm_debug_phase_ts)
  int pb = 0, poy = 0, pox = 0;
  if constexpr (CONV && SCHED != 2) {
    const int gr0 = m0 + a_base_row(0) + lrow, hw = g.cHout * g.cWout;
    pb = gr0 / hw;
    const int rem = gr0 - pb * hw;
    poy = rem / g.cWout; pox = rem - poy * g.cWout;
  }
#pragma unroll
  for (int j = 0; j < A_GROUPS; ++j) {
    const int tr_ = a_base_row(j) + lrow;                          // row inside the tile
    const int lchunk = lslot ^ ring_swz<BK>(tr_);
    const int gr = m0 + tr_;
    if constexpr (CONV) {
      int b, oy, ox;
      if constexpr (SCHED != 2) {
        b = pb; oy = poy; ox = pox;
        pox += RPI;                                                // -> row of instruction j + 1
        while (pox >= g.cWout) { pox -= g.cWout; if (++poy == g.cHout) { poy = 0; ++pb; } }
      } else {
        const int hw = g.cHout * g.cWout;
        b = gr / hw;
        const int rem = gr - b * hw;
        oy = rem / g.cWout; ox = rem - oy * g.cWout;
      }
      arow[j] = A + (long)b * g.cImg + lchunk * 8;
      ciy[j] = (gr < g.M) ? oy * g.cStride - g.cPad : -(1 << 28);
      cix[j] = ox * g.cStride - g.cPad;
    } else if constexpr (STEM) {
      const int hw = g.cHout * g.cWout;
      const int gr2 = gr < g.M ? gr : g.M - 1;
      const int b = gr2 / hw, rem = gr2 - b * hw;
      const int oy = rem / g.cWout, ox = rem - oy * g.cWout;
      arow[j] = A + (long)b * g.cImg + ((long)(2 * oy + (lchunk >> 2)) * g.cW + 2 * ox) * 4 + (lchunk & 3) * 8;
      ciy[j] = 0; cix[j] = 0;
    } else {
      arow[j] = (gr < g.M) ? A + (long)gr * g.lda + lchunk * 8 : nullptr;
    }
  }
  // (SCHED = 2 takes ONE column tile per workgroup: with the row-panel loop around it the epilogue's conditional bias loads are
  //  still "pending" in the compiler's model at the back edge and it puts an s_waitcnt vmcnt(0) in front of the first ds_read
  //  that reuses their registers -- inside the k-loop, draining the counted prefetch once per slab)
  const int n_loop = SCHED == 2 ? 1 : g.n_loop;
  for (int nt = 0; nt < n_loop; ++nt) {
  const int n0 = (bx * n_loop + nt) * BN;
  if (n0 >= g.N) break;
  if (nt > 0) __syncthreads();
#pragma unroll
  for (int j = 0; j < B_GROUPS; ++j) {
    const int tr_ = b_base_row(j) + lrow;
    const int gr = n0 + tr_;
    brow[j] = (gr < g.N) ? W + (long)gr * g.ldw + (lslot ^ ring_swz<BK>(tr_)) * 8 : nullptr;
  }
  auto stage = [&](int kt, int buf) {
    if constexpr (ABLATE == 2) { if (kt > 0) return; }
    int k0 = kt * BK;
    int tr = 0, ts = 0, ic0 = k0;
    if constexpr (CONV) {
      int tap;
      if (g.korder) { const int nt_ = g.cR * g.cS, ch = kt / nt_; tap = kt - ch * nt_; ic0 = ch * BK; k0 = tap * g.cCin + ic0; }
      else { tap = k0 / g.cCin; ic0 = k0 - tap * g.cCin; }
      tr = tap / g.cS; ts = tap - tr * g.cS;
    }
    unsigned char* la = lds + buf * STAGE + wave * (A_GROUPS * 1024);
    unsigned char* lb = lds + buf * STAGE + BM * ROWB + wave * (B_GROUPS * 1024);
#pragma unroll
    for (int j = 0; j < A_GROUPS; ++j) {
      const void* src;
      if constexpr (CONV) {
        const int iy = ciy[j] + tr * g.cDil, ix = cix[j] + ts * g.cDil;
        const bool ok = (iy >= 0) && (iy < g.cH) && (ix >= 0) && (ix < g.cW);
        src = ok ? (const void*)(arow[j] + ((long)iy * g.cW + ix) * g.cPix + ic0) : (const void*)g_zero16;
      } else if constexpr (STEM) {
        src = (const void*)(arow[j] + (long)(2 * kt) * g.cW * 4);
      } else {
        src = arow[j] ? (const void*)(arow[j] + k0) : (const void*)g_zero16;
      }
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(la + j * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_GROUPS; ++j) {
      const void* src = brow[j] ? (const void*)(brow[j] + k0) : (const void*)g_zero16;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(lb + j * 1024), 16, 0, 0);
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](int buf) {
    if constexpr (ABLATE == 1) { asm volatile("" ::"v"(buf)); return; }
    const unsigned char* la = lds + buf * STAGE;
    const unsigned char* lb = la + BM * ROWB;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      bf16x8 af[TM], bfr[TN];
      const int ch = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * (BM / WM) + i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(la + row * ROWB + ((ch ^ ring_swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * (BN / WN) + j * 32 + (lane & 31);
        bfr[j] = *(const bf16x8*)(lb + row * ROWB + ((ch ^ ring_swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
  };

  constexpr int VEC = 16 / sizeof(TOUT);
  constexpr int TPR = BN / VEC;
  constexpr int RPP = NT / TPR;
  constexpr int NP = (BAND + RPP - 1) / RPP;
  const int tcol = (tid % TPR) * VEC, trow = tid / TPR;
  const int n = n0 + tcol;
  const bool vec_ok = ((g.ldc % VEC) == 0) && ((((size_t)C) & 15) == 0) && (!R || (((size_t)R) & 15) == 0);
  auto out_row = [&](int i, int p) {
    const int brow_i = p * RPP + trow;
    if (brow_i >= BAND) return -1;
    const int m = m0 + (brow_i >> 5) * (BM / WM) + i * 32 + (brow_i & 31);
    return (m < g.M && n < g.N) ? m : -1;
  };
  // register epilogue: this lane's row / first column of tile (i, j), 16-byte piece p
  const int ehalf = lane >> 5;
  auto erow = [&](int i) { return m0 + wr * (BM / WM) + i * 32 + (lane & 31); };
  auto ecol = [&](int j, int p) { return n0 + wc * (BN / WN) + j * 32 + 16 * p + 8 * ehalf; };
  constexpr bool RDIR = RESID && EPI == 1 && sizeof(TOUT) == 2;
  uint4 rdir[RDIR ? TM : 1][RDIR ? TN : 1][2];
  if constexpr (RDIR) if (R && vec_ok) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int m = erow(i), c = ecol(j, p);
          rdir[i][j][p] = (m < g.M && c + 8 <= g.N) ? *(const uint4*)(R + (long)m * g.ldc + c) : make_uint4(0, 0, 0, 0);
        }
  }
  uint4 rpre[(RESID && EPI == 0) ? TM : 1][(RESID && EPI == 0) ? NP : 1];
  if constexpr (RESID && EPI == 0) if (R && vec_ok) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int m = out_row(i, p);
        rpre[i][p] = (m >= 0 && n + VEC <= g.N) ? *(const uint4*)(R + (long)m * g.ldc + n) : make_uint4(0, 0, 0, 0);
      }
  }

  const int nk = g.K / BK;
  if constexpr (SCHED == 0) {
    constexpr int D = NSTAGE - 1;                        // slabs in flight
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < nk) stage(s, s);
    for (int kt = 0; kt < nk; ++kt) {
      // slabs issued so far: min(kt + D, nk); everything up to slab kt must have landed
      const int ahead = (kt + D < nk ? kt + D : nk) - kt - 1;          // 0 .. D-1 (wave uniform)
      if (ahead >= 2) wait_vm_barrier<2 * LPS>();
      else if (ahead == 1) wait_vm_barrier<LPS>();
      else wait_vm_barrier<0>();
      if (kt + D < nk) stage(kt + D, (kt + D) % NSTAGE);
      compute(kt % NSTAGE);
    }
  } else if constexpr (SCHED == 5) {
    static_assert(SCHED != 5 || (BM == 256 && BN == 256 && WM == 2 && WN == 4 && BK == 64 && NSTAGE == 2 && !RESID && !STEM && EPI == 0 && ABLATE == 0),
                  "hand-scheduled k-loop: 256 x 256 x 64 ring tile, 2 x 4 waves, no shortcut operand");
    // (host side, launch_bf16 case 18: N % 256 == 0, K % 64 == 0, taps <= 32, filter offsets < 2^32, korder for R * S > 1)
    const unsigned lds0 = (unsigned)(unsigned long)(las_ptr)lds;
    const int l31 = lane & 31, hf = lane >> 5, sw = (lane >> 1) & 7;          // ring_swz<64>(row) depends on the lane only: rows are 32-aligned per fragment
    const unsigned rowA = lds0 + (unsigned)((wr * (BM / WM) + l31) * ROWB), rowB = lds0 + (unsigned)((wc * (BN / WN) + l31) * ROWB);
    const unsigned c0 = (unsigned)(((0 + hf) ^ sw) << 4), c1 = (unsigned)(((2 + hf) ^ sw) << 4);   // k-steps 0 / 1; 2 / 3 are these ^ 0x40
    // per-lane sources of the four activation loads: pixel pointer of tap (0, 0) + validity bit per tap
    const unsigned short* ab[A_GROUPS];
    unsigned mk[A_GROUPS];
#pragma unroll
    for (int j = 0; j < A_GROUPS; ++j) {
      if constexpr (CONV) {
        ab[j] = arow[j] + ((long)ciy[j] * g.cW + cix[j]) * g.cPix;
        unsigned cm = 0, m = 0;                              // tap (tr, ts) in bounds <=> row tr and column ts are: column bits, replicated per valid row
        for (int ts = 0; ts < g.cS; ++ts) { const int ix = cix[j] + ts * g.cDil; if (ix >= 0 && ix < g.cW) cm |= 1u << ts; }
        for (int tr = 0; tr < g.cR; ++tr) { const int iy = ciy[j] + tr * g.cDil; if (iy >= 0 && iy < g.cH) m |= cm << (tr * g.cS); }
        mk[j] = m;
      } else {
        ab[j] = arow[j] ? arow[j] : A;
        mk[j] = arow[j] ? 1u : 0u;
      }
    }
    unsigned wo[B_GROUPS];
#pragma unroll
    for (int j = 0; j < B_GROUPS; ++j) {
      const int tr_ = b_base_row(j) + lrow;
      wo[j] = (unsigned)((((long)(n0 + tr_) * g.ldw) + (lslot ^ ring_swz<BK>(tr_)) * 8) * 2);
    }
    const unsigned zlo = (unsigned)(unsigned long)(const void*)g_zero16, zhi = (unsigned)((unsigned long)(const void*)g_zero16 >> 32);
    // wave-uniform part of a slab's sources, stepped from slab to slab with scalar adds (k order = (channel chunk, tap); the host
    // admits only that order or single-tap layers): byte offset added to every activation pointer, filter pointer, tap bit
    const int q_ntap = CONV ? g.cR * g.cS : 1, q_S = CONV ? g.cS : 1;
    const long q_dp2 = CONV ? (long)g.cDil * g.cPix * 2 : 0;                         // one tap to the right
    const long q_row = CONV ? ((long)g.cW - (q_S - 1)) * q_dp2 : 0;                   // last tap of a filter row -> first tap of the next
    const long q_back = (long)BK * 2 - (CONV ? ((long)(g.cR - 1) * g.cW + (q_S - 1)) * q_dp2 : 0);   // last tap -> tap 0 of the next chunk
    const long q_wtap = CONV ? g.cCin : 0, q_wback = (long)BK - (long)(q_ntap - 1) * q_wtap;
    int q_tap = 0, q_ts = 0;
    long so = 0; const unsigned short* wb = W; unsigned tb = 1u;
    auto slab_next = [&]() {
      if (++q_tap == q_ntap) { q_tap = 0; q_ts = 0; tb = 1u; so += q_back; wb += q_wback; }
      else { tb <<= 1; wb += q_wtap; if (++q_ts == q_S) { q_ts = 0; so += q_row; } else so += q_dp2; }
    };
    stage(0, 0);
    if (g.phase_ts) ts_loop = wall_clock64();
    agpr_zero128();
    for (int kt = 0; kt + 1 < nk; ++kt) {
      const unsigned cur = (unsigned)(kt & 1) * STAGE, nxt = STAGE - cur;
      const unsigned la0 = rowA + cur + c0, la1 = rowA + cur + c1, lb0 = rowB + cur + c0, lb1 = rowB + cur + c1;
      const unsigned ma = __builtin_amdgcn_readfirstlane(lds0 + nxt + wave * (A_GROUPS * 1024));
      const unsigned mb = __builtin_amdgcn_readfirstlane(lds0 + nxt + BM * ROWB + wave * (B_GROUPS * 1024));
      slab_next();
      asm volatile(RSLAB_LOAD
                   :
                   : [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1),
                     [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [ab3] "v"(ab[3]),
                     [mk0] "v"(mk[0]), [mk1] "v"(mk[1]), [mk2] "v"(mk[2]), [mk3] "v"(mk[3]),
                     [wo0] "v"(wo[0]), [wo1] "v"(wo[1]), [wo2] "v"(wo[2]), [wo3] "v"(wo[3]),
                     [zlo] "v"(zlo), [zhi] "v"(zhi), [so] "s"(so), [wb] "s"(wb), [tb] "s"(tb), [ma] "s"(ma), [mb] "s"(mb)
                   : "memory", "vcc", "scc", "m0", RCL_VTMP, RCL_AGPR);
    }
    {
      const unsigned cur = (unsigned)((nk - 1) & 1) * STAGE;
      const unsigned la0 = rowA + cur + c0, la1 = rowA + cur + c1, lb0 = rowB + cur + c0, lb1 = rowB + cur + c1;
      asm volatile(RSLAB_LAST
                   :
                   : [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1)
                   : "memory", RCL_VTMP, RCL_AGPR);
    }
    // (the accumulators stay in a[0:127]; the epilogue below fetches them one 32-row band at a time -- reading all 128 at once
    //  made the compiler park some of the values in "free" AGPRs, i.e. in accumulators that had not been read yet)
  } else if constexpr (SCHED == 6) {
    static_assert(SCHED != 6 || (BM == 256 && BN == 256 && WM == 2 && WN == 4 && BK == 64 && NSTAGE == 2 && !RESID && !STEM && EPI == 0 && ABLATE == 0),
                  "asymmetric-ring k-loop: 256 x 256 x 64 tile, 2 x 4 waves, no shortcut operand");
    constexpr unsigned SLOT = BM * ROWB, WBASE = 3 * SLOT;         // activation slots 0..2 at 0, filter slots 0..1 at 96 KB
    const unsigned lds0 = (unsigned)(unsigned long)(las_ptr)lds;
    const int l31 = lane & 31, hf = lane >> 5, sw = (lane >> 1) & 7;
    const unsigned rowA = lds0 + (unsigned)((wr * (BM / WM) + l31) * ROWB), rowB = lds0 + WBASE + (unsigned)((wc * (BN / WN) + l31) * ROWB);
    const unsigned c0 = (unsigned)(((0 + hf) ^ sw) << 4), c1 = (unsigned)(((2 + hf) ^ sw) << 4);
    const unsigned short* ab[A_GROUPS];
    unsigned mk[A_GROUPS];
#pragma unroll
    for (int j = 0; j < A_GROUPS; ++j) {
      if constexpr (CONV) {
        ab[j] = arow[j] + ((long)ciy[j] * g.cW + cix[j]) * g.cPix;
        unsigned cm = 0, m = 0;                              // tap (tr, ts) in bounds <=> row tr and column ts are: column bits, replicated per valid row
        for (int ts = 0; ts < g.cS; ++ts) { const int ix = cix[j] + ts * g.cDil; if (ix >= 0 && ix < g.cW) cm |= 1u << ts; }
        for (int tr = 0; tr < g.cR; ++tr) { const int iy = ciy[j] + tr * g.cDil; if (iy >= 0 && iy < g.cH) m |= cm << (tr * g.cS); }
        mk[j] = m;
      } else {
        ab[j] = arow[j] ? arow[j] : A;
        mk[j] = arow[j] ? 1u : 0u;
      }
    }
    unsigned wo[B_GROUPS];
#pragma unroll
    for (int j = 0; j < B_GROUPS; ++j) {
      const int tr_ = b_base_row(j) + lrow;
      wo[j] = (unsigned)((((long)(n0 + tr_) * g.ldw) + (lslot ^ ring_swz<BK>(tr_)) * 8) * 2);
    }
    const unsigned zlo = (unsigned)(unsigned long)(const void*)g_zero16, zhi = (unsigned)((unsigned long)(const void*)g_zero16 >> 32);
    // two scalar steppers over the k-slabs (k order = (channel chunk, tap)): the activation side runs one slab ahead of the filter side
    const int q_ntap = CONV ? g.cR * g.cS : 1, q_S = CONV ? g.cS : 1;
    const long q_dp2 = CONV ? (long)g.cDil * g.cPix * 2 : 0;
    const long q_row = CONV ? ((long)g.cW - (q_S - 1)) * q_dp2 : 0;
    const long q_back = (long)BK * 2 - (CONV ? ((long)(g.cR - 1) * g.cW + (q_S - 1)) * q_dp2 : 0);
    const long q_wtap = CONV ? g.cCin : 0, q_wback = (long)BK - (long)(q_ntap - 1) * q_wtap;
    // (plain statements, not mutating lambdas: with by-reference captures the compiler kept this state in scratch memory and put
    //  a vmcnt(0) for its reloads into the k-loop, draining the counted LDS-DMA pipeline)
    int a_tap = 0, a_ts = 0, w_tap = 0;
    long so = 0; unsigned tb = 1u; const unsigned short* wb = W;
#define RELNET_A_NEXT()                                                                      \
    if (++a_tap == q_ntap) { a_tap = 0; a_ts = 0; tb = 1u; so += q_back; }                   \
    else { tb <<= 1; if (++a_ts == q_S) { a_ts = 0; so += q_row; } else so += q_dp2; }
#define RELNET_W_NEXT() if (++w_tap == q_ntap) { w_tap = 0; wb += q_wback; } else wb += q_wtap;
    // prologue (compiler-issued LDS-direct loads, in the order the counted waits assume): A(0), W(0), A(1)
#pragma unroll
    for (int j = 0; j < A_GROUPS; ++j) {
      const void* src = (mk[j] & 1u) ? (const void*)ab[j] : (const void*)g_zero16;
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(lds + (wave * A_GROUPS + j) * 1024), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < B_GROUPS; ++j)
      __builtin_amdgcn_global_load_lds((gas_ptr)((const char*)W + wo[j]), (las_ptr)(lds + WBASE + (wave * B_GROUPS + j) * 1024), 16, 0, 0);
    if (nk > 1) {
      RELNET_A_NEXT()
#pragma unroll
      for (int j = 0; j < A_GROUPS; ++j) {
        const void* src = (mk[j] & tb) ? (const void*)((const char*)ab[j] + so) : (const void*)g_zero16;
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(lds + SLOT + (wave * A_GROUPS + j) * 1024), 16, 0, 0);
      }
    }
    if (g.phase_ts) ts_loop = wall_clock64();
    agpr_zero128();
    unsigned aslot = 0;                                   // kt % 3
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned la0 = rowA + aslot * SLOT + c0, la1 = rowA + aslot * SLOT + c1;
      const unsigned lb0 = rowB + (unsigned)(kt & 1) * SLOT + c0, lb1 = rowB + (unsigned)(kt & 1) * SLOT + c1;
      const unsigned aslot2 = aslot == 0 ? 2u : aslot - 1u;                        // (kt + 2) % 3
      if (kt + 2 < nk) {
        RELNET_A_NEXT() RELNET_W_NEXT()                   // activation slab kt + 2, filter slab kt + 1
        const unsigned ma = __builtin_amdgcn_readfirstlane(lds0 + aslot2 * SLOT + wave * (A_GROUPS * 1024));
        const unsigned mb = __builtin_amdgcn_readfirstlane(lds0 + WBASE + (unsigned)((kt + 1) & 1) * SLOT + wave * (B_GROUPS * 1024));
        asm volatile(RSLAB6_FULL
                     :
                     : [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1),
                       [ab0] "v"(ab[0]), [ab1] "v"(ab[1]), [ab2] "v"(ab[2]), [ab3] "v"(ab[3]),
                       [mk0] "v"(mk[0]), [mk1] "v"(mk[1]), [mk2] "v"(mk[2]), [mk3] "v"(mk[3]),
                       [wo0] "v"(wo[0]), [wo1] "v"(wo[1]), [wo2] "v"(wo[2]), [wo3] "v"(wo[3]),
                       [zlo] "v"(zlo), [zhi] "v"(zhi), [so] "s"(uniform64((unsigned long)so)), [wb] "s"(uniform64((unsigned long)wb)), [tb] "s"(__builtin_amdgcn_readfirstlane(tb)), [ma] "s"(ma), [mb] "s"(mb)
                     : "memory", "vcc", "scc", "m0", RCL_VTMP, RCL_AGPR);
      } else if (kt + 1 < nk) {
        RELNET_W_NEXT()
        const unsigned mb = __builtin_amdgcn_readfirstlane(lds0 + WBASE + (unsigned)((kt + 1) & 1) * SLOT + wave * (B_GROUPS * 1024));
        asm volatile(RSLAB6_WONLY
                     :
                     : [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1),
                       [wo0] "v"(wo[0]), [wo1] "v"(wo[1]), [wo2] "v"(wo[2]), [wo3] "v"(wo[3]), [wb] "s"(uniform64((unsigned long)wb)), [mb] "s"(mb)
                     : "memory", "scc", "m0", RCL_VTMP, RCL_AGPR);
      } else {
        asm volatile(RSLAB6_LAST
                     :
                     : [la0] "v"(la0), [la1] "v"(la1), [lb0] "v"(lb0), [lb1] "v"(lb1)
                     : "memory", RCL_VTMP, RCL_AGPR);
      }
      aslot = aslot == 2 ? 0u : aslot + 1u;
    }
#undef RELNET_A_NEXT
#undef RELNET_W_NEXT
  } else if constexpr (SCHED == 4) {
    // Window schedule for 3x3 / stride 1 layers (host checks the geometry): the k-loop runs (channel chunk, tap); the nine taps of a
    // 64-channel chunk read the SAME input pixels shifted by ((r - 1) W + (s - 1)) dil, so the A operand of a chunk is staged ONCE
    // as a window of the flattened pixel axis -- rows [m0 - halo, m0 + 256 + halo), halo = (W + 1) dil -- and every tap reads its
    // fragments from the window at a row offset; lanes whose tap leaves the image take zeros (a 9-bit mask per fragment row).
    // L2 -> LDS fill per chunk: 48 KB window + 9 x 32 KB filter slabs instead of 9 x 64 KB.
    // LDS: window [2][384 rows][128 B] (double buffered over chunks) + filter slab [2][256][128 B] = 160 KB.
    static_assert(SCHED != 4 || (BM == 256 && BN == 256 && WM == 2 && WN == 4 && BK == 64 && NSTAGE == 2 && !RESID && MODE == 1), "window schedule: 256 x 256 x 64 conv tile");
    constexpr int WROWS = 384, WIN_BYTES = WROWS * ROWB, WSLAB = BN * ROWB;
    unsigned char* const win = lds;
    unsigned char* const wsl = lds + 2 * WIN_BYTES;
    const int halo = (g.cW + 1) * g.cDil;
    const int hw = g.cH * g.cW;
    unsigned wsrc[6], wofs[B_GROUPS];                                  // 32-bit element offsets (the maps are < 2^31 elements: host check)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int w = (wave * 6 + j) * RPI + lrow;
      long gp = (long)m0 - halo + w;
      gp = gp < 0 ? 0 : (gp >= g.M ? (long)g.M - 1 : gp);              // rows outside the tensor are only ever read by masked taps
      const int b = (int)(gp / hw), rem = (int)(gp - (long)b * hw);
      wsrc[j] = (unsigned)((long)b * g.cImg + (long)rem * g.cPix + ((lslot ^ ring_swz<BK>(w)) * 8));
    }
#pragma unroll
    for (int j = 0; j < B_GROUPS; ++j) {
      const int tr_ = b_base_row(j) + lrow;
      wofs[j] = (n0 + tr_ < g.N) ? (unsigned)((long)(n0 + tr_) * g.ldw + (lslot ^ ring_swz<BK>(tr_)) * 8) : 0xffffffffu;
    }
    unsigned vmask[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int gr = m0 + wr * (BM / WM) + i * 32 + (lane & 31);
      // branch free: 3 row bits x 3 column bits -> 9 tap bits (the per-bit select form spilled 36 registers to scratch)
      const int grc = gr < g.M ? gr : g.M - 1;
      const int rem = grc % hw, oy = rem / g.cW, ox = rem - oy * g.cW;
      const unsigned my = (unsigned)(oy - g.cDil >= 0) | 2u | ((unsigned)(oy + g.cDil < g.cH) << 2);
      const unsigned mx = (unsigned)(ox - g.cDil >= 0) | 2u | ((unsigned)(ox + g.cDil < g.cW) << 2);
      const unsigned mk = (0u - (my & 1u)) & mx | ((0u - ((my >> 1) & 1u)) & (mx << 3)) | ((0u - ((my >> 2) & 1u)) & (mx << 6));
      vmask[i] = gr < g.M ? mk : 0u;
    }
    auto stage_win = [&](int chunk, int wb) {
#pragma unroll
      for (int j = 0; j < 6; ++j)
        __builtin_amdgcn_global_load_lds((gas_ptr)(A + wsrc[j] + chunk * BK), (las_ptr)(win + wb * WIN_BYTES + (wave * 6 + j) * 1024), 16, 0, 0);
    };
    auto stage_w = [&](int chunk, int tap, int buf) {
      const int k0 = tap * g.cCin + chunk * BK;
#pragma unroll
      for (int j = 0; j < B_GROUPS; ++j) {
        const void* src = wofs[j] != 0xffffffffu ? (const void*)(W + wofs[j] + k0) : (const void*)g_zero16;
        __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(wsl + buf * WSLAB + (wave * B_GROUPS + j) * 1024), 16, 0, 0);
      }
    };
    auto compute_win = [&](int buf, int wb, int tap) {
      const int tr = tap / 3, ts = tap - 3 * tr;
      const int toff = ((tr - 1) * g.cW + (ts - 1)) * g.cDil + halo;
      const unsigned char* wp = win + wb * WIN_BYTES;
      const unsigned char* lb = wsl + buf * WSLAB;
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
        bf16x8 af[TM], bfr[TN];
        const int ch = 2 * kk + (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wr * (BM / WM) + i * 32 + (lane & 31) + toff;
          const bf16x8 v = *(const bf16x8*)(wp + row * ROWB + ((ch ^ ring_swz<BK>(row)) << 4));
          const short m = (short)-(int)((vmask[i] >> tap) & 1u);
          af[i] = v & m;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wc * (BN / WN) + j * 32 + (lane & 31);
          bfr[j] = *(const bf16x8*)(lb + row * ROWB + ((ch ^ ring_swz<BK>(row)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    };
    const int nchunk = g.cCin / BK;
    stage_win(0, 0);
    stage_w(0, 0, 0);
    int kt = 0;
#pragma unroll 1
    for (int chunk = 0; chunk < nchunk; ++chunk) {
#pragma unroll 1
      for (int tap = 0; tap < 9; ++tap, ++kt) {      // (not unrolled: the per-tap fragment addresses are loop invariant over the chunks and
                                                     //  would be hoisted into ~40 live registers -> scratch spills)
        wait_vm_barrier<0>();                 // this slab (and, at tap 0, this chunk's window) has landed; every wave is past slab kt - 1
        if (tap < 8) stage_w(chunk, tap + 1, (kt + 1) & 1);
        else if (chunk + 1 < nchunk) stage_w(chunk + 1, 0, (kt + 1) & 1);
        if (tap == 0 && chunk + 1 < nchunk) stage_win(chunk + 1, (chunk + 1) & 1);   // that buffer was last read in chunk - 1
        compute_win(kt & 1, chunk & 1, tap);
      }
    }
  } else if constexpr (SCHED == 2) {
    // Ping-pong schedule (the guide's 256 x 256 "8-phase" structure, rebuilt here for 32x32x16 MFMAs and implicit-GEMM A rows).
    // A k-slab is consumed in phases of one or two quadrants (64 x 32 of the wave's 128 x 64 outputs, 8 MFMAs each); a phase is
    //   L: ds_read the register fragments it needs, issue its share of the prefetch (LDS-direct loads of half tiles), s_waitcnt,
    //   s_barrier;  M: the MFMAs at s_setprio 1;  s_barrier
    // and the two wave rows (wr = 0 / 1: one wave of each on every SIMD) run half a phase apart (wr = 1 takes one extra barrier
    // up front), so on every SIMD one wave is in M while the other is in L: LDS reads, address arithmetic and load issue hide
    // behind the other wave's MFMAs instead of in front of this wave's own.
    //   staging order S(n): slab n >> 2, kind n & 3 in {A half 0, W half 0, W half 1, A half 1}; the LDS holds two whole slabs
    //   and 3..4 half tiles are in flight at every wait (counted vmcnt, never 0 in steady state);
    //   WAR: a slot is re-staged one phase after its last read at the earliest; every phase waits lgkmcnt(0) BEFORE its barrier,
    //        so those reads have returned on both wave rows before any wave issues the next phase's stage.
    static_assert(BM == 256 && BN == 256 && WM == 2 && WN == 4 && BK == 64 && NSTAGE == 2 && !RESID, "ping-pong schedule: 256 x 256 x 64, 2 x 4 waves");
    bf16x8 fa[2][4], fb[2][4];                            // A half in use [row fragment][k-step]; both W halves [half][k-step]
    bool pp_loop = false;                                 // (ablation only)
    auto stage_half = [&](int t, auto kindc) {             // slab t, kind: 0 = A0, 1 = W0, 2 = W1, 3 = A1
      constexpr int KIND = decltype(kindc)::value;
      if (t >= nk) return;
      if constexpr (ABLATE == 2) { if (pp_loop) return; }
      int k0 = t * BK, tap_ = 0, ic0 = k0;
      if constexpr (CONV) {
        if (g.korder) { const int nt_ = g.cR * g.cS, ch = t / nt_; tap_ = t - ch * nt_; ic0 = ch * BK; k0 = tap_ * g.cCin + ic0; }
        else { tap_ = k0 / g.cCin; ic0 = k0 - tap_ * g.cCin; }
      }
      const unsigned lbase = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(las_ptr)lds + (t & 1) * STAGE);
      if constexpr (KIND == 0 || KIND == 3) {
        constexpr int H = KIND == 3 ? 1 : 0;
        int tr = 0, ts = 0;
        if constexpr (CONV) { tr = tap_ / g.cS; ts = tap_ - tr * g.cS; }
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int j = 2 * H + ii;
          const void* src;
          if constexpr (CONV) {
            const int iy = ciy[j] + tr * g.cDil, ix = cix[j] + ts * g.cDil;
            const bool ok = (iy >= 0) && (iy < g.cH) && (ix >= 0) && (ix < g.cW);
            src = ok ? (const void*)(arow[j] + ((long)iy * g.cW + ix) * g.cPix + ic0) : (const void*)g_zero16;
          } else if constexpr (STEM) {
            src = (const void*)(arow[j] + (long)(2 * t) * g.cW * 4);
          } else {
            src = arow[j] ? (const void*)(arow[j] + k0) : (const void*)g_zero16;
          }
          glds16_asm(src, lbase + a_base_row(j) * ROWB);
        }
      } else {
        constexpr int H = KIND - 1;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int j = 2 * H + ii;
          const void* src = brow[j] ? (const void*)(brow[j] + k0) : (const void*)g_zero16;
          glds16_asm(src, lbase + BM * ROWB + b_base_row(j) * ROWB);
        }
      }
    };
    auto read_a = [&](int t, int h) {
      if constexpr (ABLATE == 1) return;
      const unsigned char* la = lds + (t & 1) * STAGE;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        const int row = wr * 128 + h * 64 + i2 * 32 + (lane & 31);
        const unsigned char* rp = la + row * ROWB;
        const int sw = ring_swz<BK>(row);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) fa[i2][kk] = *(const bf16x8*)(rp + (((2 * kk + (lane >> 5)) ^ sw) << 4));
      }
    };
    auto read_b = [&](int t, int h) {
      if constexpr (ABLATE == 1) return;
      const unsigned char* lb = lds + (t & 1) * STAGE + BM * ROWB;
      const int row = wc * 64 + h * 32 + (lane & 31);
      const unsigned char* rp = lb + row * ROWB;
      const int sw = ring_swz<BK>(row);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fb[h][kk] = *(const bf16x8*)(rp + (((2 * kk + (lane >> 5)) ^ sw) << 4));
    };
    auto quad = [&](auto ac, auto bc) {
      constexpr int QA = decltype(ac)::value, QB = decltype(bc)::value;
      if constexpr (ABLATE == 1) return;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
          acc[2 * QA + i2][QB] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[QB][kk], fa[i2][kk], acc[2 * QA + i2][QB], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
    };
    using k0c = std::integral_constant<int, 0>;
    using k1c = std::integral_constant<int, 1>;
    using k2c = std::integral_constant<int, 2>;
    using k3c = std::integral_constant<int, 3>;
#define RELNET_PP_SYNC(WAIT)                                           \
    __builtin_amdgcn_sched_barrier(0);                                 \
    asm volatile(WAIT "\n\ts_barrier" ::: "memory");                   \
    __builtin_amdgcn_sched_barrier(0);
#define RELNET_PP_BAR()                                                \
    __builtin_amdgcn_sched_barrier(0);                                 \
    asm volatile("s_barrier" ::: "memory");                            \
    __builtin_amdgcn_sched_barrier(0);
    // TWO phases per slab (16 MFMAs = 512 MFMA-pipe cycles each, against ~16 ds_reads + LDS latency of the other wave row's L part;
    // the first version ran four 8-MFMA phases and its L parts were LONGER than the M parts: PMC, profiles/r03_conv3x3_pmc.json):
    //   phase 2t    : L = A half 0 + both W halves of slab t (16 reads), stage S(4t + 6), S(4t + 7); M = quadrants (0,0), (0,1)
    //   phase 2t + 1: L = A half 1 (8 reads),                           stage S(4t + 8), S(4t + 9); M = quadrants (1,0), (1,1)
    // i.e. phase p stages S(2p + 6), S(2p + 7) into the slots of S(2p - 2), S(2p - 1), last read in phase p - 1 at the latest.
    // RAW: even phases wait vmcnt(8) (A half 1 of this slab = S(4t + 3) must have landed, the four younger half tiles may fly),
    //      odd phases vmcnt(6) (slab t + 1's A0 / W0 / W1 = S(4t + 4 .. 6) landed, three younger in flight).
    stage_half(0, k0c{}); stage_half(0, k1c{}); stage_half(0, k2c{}); stage_half(0, k3c{});
    stage_half(1, k0c{}); stage_half(1, k1c{});
    if (nk >= 2) { RELNET_PP_SYNC("s_waitcnt vmcnt(6)") } else { RELNET_PP_SYNC("s_waitcnt vmcnt(0)") }
    if (wr == 1) { RELNET_PP_BAR() }
    pp_loop = true;
    for (int t = 0; t < nk; ++t) {
      read_a(t, 0); read_b(t, 0); read_b(t, 1);
      stage_half(t + 1, k2c{}); stage_half(t + 1, k3c{});
      if (t + 2 <= nk) { RELNET_PP_SYNC("s_waitcnt vmcnt(8) lgkmcnt(0)") } else { RELNET_PP_SYNC("s_waitcnt vmcnt(0) lgkmcnt(0)") }
      quad(k0c{}, k0c{});
      quad(k0c{}, k1c{});
      RELNET_PP_BAR()
      read_a(t, 1);
      stage_half(t + 2, k0c{}); stage_half(t + 2, k1c{});
      if (t + 3 <= nk) { RELNET_PP_SYNC("s_waitcnt vmcnt(6) lgkmcnt(0)") } else { RELNET_PP_SYNC("s_waitcnt vmcnt(0) lgkmcnt(0)") }
      quad(k1c{}, k0c{});
      quad(k1c{}, k1c{});
      RELNET_PP_BAR()
    }
    if (wr == 0) { RELNET_PP_BAR() }
#undef RELNET_PP_SYNC
#undef RELNET_PP_BAR
  } else {
    constexpr int KK = BK / 16;
    static_assert(KK % 2 == 0, "two fragment sets alternate per k-step");
    bf16x8 fa[2][TM], fb[2][TN];
    auto load_frag = [&](int buf, int kk, bf16x8 (&xa)[TM], bf16x8 (&xb)[TN]) {
      if constexpr (ABLATE == 1) return;
      const unsigned char* la = lds + buf * STAGE;
      const unsigned char* lb = la + BM * ROWB;
      const int ch = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * (BM / WM) + i * 32 + (lane & 31);
        xa[i] = *(const bf16x8*)(la + row * ROWB + ((ch ^ ring_swz<BK>(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * (BN / WN) + j * 32 + (lane & 31);
        xb[j] = *(const bf16x8*)(lb + row * ROWB + ((ch ^ ring_swz<BK>(row)) << 4));
      }
    };
    // MFMAs of one k-step, rows [I0, I1) of the wave's TM x TN tile grid
    auto mma = [&](const bf16x8 (&xa)[TM], const bf16x8 (&xb)[TN], auto i0c, auto i1c) {
      constexpr int I0 = decltype(i0c)::value, I1 = decltype(i1c)::value;
      if constexpr (ABLATE == 1) return;
#pragma unroll
      for (int i = I0; i < I1; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[j], xa[i], acc[i][j], 0, 0, 0);
    };
    using c0 = std::integral_constant<int, 0>;
    using c1 = std::integral_constant<int, 1>;
    using cT = std::integral_constant<int, TM>;
    // one k-step: the first row of MFMAs consumes fragments that landed a whole step ago (the compiler's lgkmcnt(0) in
    // front of them is then free), the reads of the NEXT fragments are issued behind it and have the remaining
    // (TM-1) x TN MFMAs to land
#define RELNET_KSTEP(CUR, NEXT_LOAD)                 \
    mma(fa[CUR], fb[CUR], c0{}, c1{});               \
    __builtin_amdgcn_sched_barrier(0);               \
    NEXT_LOAD;                                       \
    __builtin_amdgcn_sched_barrier(0);               \
    mma(fa[CUR], fb[CUR], c1{}, cT{});               \
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
      if (s < nk) stage(s, s);
    {
      const int ahead = (NSTAGE < nk ? NSTAGE : nk) - 1;               // slabs that may still be in flight
      if (ahead >= 3) wait_vm_barrier<3 * LPS>();
      else if (ahead == 2) wait_vm_barrier<2 * LPS>();
      else if (ahead == 1) wait_vm_barrier<LPS>();
      else wait_vm_barrier<0>();
    }
    load_frag(0, 0, fa[0], fb[0]);
    int kt = 0;
    // steady state (branch-free): NSTAGE - 1 slabs are in flight at every hand-over and the consumed buffer is refilled
    for (; kt + NSTAGE < nk; ++kt) {
      const int buf = kt % NSTAGE;
#pragma unroll
      for (int kk = 0; kk < KK - 1; ++kk) {
        RELNET_KSTEP(kk & 1, load_frag(buf, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]))
      }
      // hand-over: every read of this slab has returned (own lgkmcnt(0)), slab kt+1 has landed (counted vmcnt), all
      // waves agree (barrier) -> refill this buffer, fetch the first fragments of the next slab
      RELNET_KSTEP((KK - 1) & 1,
                   asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"((NSTAGE - 2) * LPS) : "memory");
                   stage(kt + NSTAGE, buf);
                   load_frag((kt + 1) % NSTAGE, 0, fa[0], fb[0]))
    }
    for (; kt < nk; ++kt) {                                            // drain: nothing left to issue
      const int buf = kt % NSTAGE;
#pragma unroll
      for (int kk = 0; kk < KK - 1; ++kk) {
        RELNET_KSTEP(kk & 1, load_frag(buf, kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]))
      }
      if (kt + 1 < nk) {
        const int ahead = nk - kt - 2;                                 // slabs beyond kt+1 still in flight: 0 .. NSTAGE-2
        RELNET_KSTEP((KK - 1) & 1,
                     if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
                     else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(LPS) : "memory");
                     else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                     load_frag((kt + 1) % NSTAGE, 0, fa[0], fb[0]))
      } else {
        mma(fa[(KK - 1) & 1], fb[(KK - 1) & 1], c0{}, cT{});
      }
    }
#undef RELNET_KSTEP
  }
  if (g.phase_ts && tid == 0) {
    long long* d = g.phase_ts + ((long)blockIdx.x + (long)gridDim.x * blockIdx.y) * 8;
    d[0] = ts_in; d[1] = ts_loop; d[2] = wall_clock64(); d[4] = (long long)__builtin_readcyclecounter() - cy_in;
  }
  if constexpr (EPI == 1) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = erow(i);
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if constexpr (sizeof(TOUT) == 2) {
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const unsigned int a = __float_as_uint(acc[i][j][8 * p + e]), b = __float_as_uint(acc[i][j][8 * p + 4 + e]);
              const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
              v[e] = __uint_as_float(r[0]); v[4 + e] = __uint_as_float(r[1]);
            }
            const int c = ecol(j, p);
            if (m >= g.M || c >= g.N) continue;
            const float brow = (g.bias_mode == 2) ? g.bias[m] : 0.f;
            TOUT* cp = C + (long)m * g.ldc + c;
            if (vec_ok && c + 8 <= g.N) {
              if (g.bias_mode == 1) {
                const float4 b0 = *(const float4*)(g.bias + c), b1 = *(const float4*)(g.bias + c + 4);
                v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
              }
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += brow;
              if constexpr (RDIR) if (R) {
                const unsigned int rw[4] = {rdir[i][j][p].x, rdir[i][j][p].y, rdir[i][j][p].z, rdir[i][j][p].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[2 * e] = res_apply(v[2 * e], bf2f(rw[e] & 0xffff), g.relu); v[2 * e + 1] = res_apply(v[2 * e + 1], bf2f(rw[e] >> 16), g.relu); }
              }
              if (g.relu == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
              }
              if constexpr (ABLATE == 3) { asm volatile("" ::"v"(v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7])); }
              else *(uint4*)cp = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            } else {
              const TOUT* rp = R ? R + (long)m * g.ldc + c : nullptr;
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                if (c + e < g.N) {
                  float x = v[e] + brow + (g.bias_mode == 1 ? g.bias[c + e] : 0.f);
                  if (rp) x = res_apply(x, load_out<TOUT>(rp + e), g.relu);
                  if (g.relu == 1) x = fmaxf(x, 0.f);
                  store_out<TOUT>(cp + e, x);
                }
              }
            }
          }
        } else {
          // fp32 outputs: a register group is already 16 contiguous bytes
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int c = n0 + wc * (BN / WN) + j * 32 + 8 * gq + 4 * ehalf;
            if (m >= g.M || c >= g.N) continue;
            const float brow = (g.bias_mode == 2) ? g.bias[m] : 0.f;
            TOUT* cp = C + (long)m * g.ldc + c;
            const TOUT* rp = R ? R + (long)m * g.ldc + c : nullptr;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * gq + e] + brow;
            if (vec_ok && c + 4 <= g.N) {
              if (g.bias_mode == 1) { const float4 b0 = *(const float4*)(g.bias + c); v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; }
              if (rp) { const float4 r0 = *(const float4*)rp; v[0] = res_apply(v[0], r0.x, g.relu); v[1] = res_apply(v[1], r0.y, g.relu); v[2] = res_apply(v[2], r0.z, g.relu); v[3] = res_apply(v[3], r0.w, g.relu); }
              if (g.relu == 1) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
              *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                if (c + e < g.N) {
                  float x = v[e] + (g.bias_mode == 1 ? g.bias[c + e] : 0.f);
                  if (rp) x = res_apply(x, load_out<TOUT>(rp + e), g.relu);
                  if (g.relu == 1) x = fmaxf(x, 0.f);
                  store_out<TOUT>(cp + e, x);
                }
              }
            }
          }
        }
      }
    }
  } else {
  __syncthreads();                                     // every wave is done reading the ring: the epilogue band reuses it

  float* ct = (float*)lds;
  float bv[VEC];
#pragma unroll
  for (int e = 0; e < VEC; ++e) bv[e] = (g.bias_mode == 1 && n + e < g.N) ? g.bias[n + e] : 0.f;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if constexpr (SCHED == 5 || SCHED == 6) {          // band i of the AGPR-resident accumulators (block i * TN + j = a[16 (2 i + j) ...])
      switch (i) {
        case 0: agpr_read16<0>(acc[0][0]); agpr_read16<16>(acc[0][1]); break;
        case 1: agpr_read16<32>(acc[1][0]); agpr_read16<48>(acc[1][1]); break;
        case 2: agpr_read16<64>(acc[2][0]); agpr_read16<80>(acc[2][1]); break;
        default: agpr_read16<96>(acc[3][0]); agpr_read16<112>(acc[3][1]); break;
      }
    }
    if (i > 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int row = wr * 32 + (lane & 31);
        const int col = wc * (BN / WN) + j * 32 + 8 * gq + 4 * (lane >> 5);
        *(float4*)(ct + row * CLD + col) = make_float4(acc[i][j][4 * gq], acc[i][j][4 * gq + 1],
                                                       acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]);
      }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int m = out_row(i, p);
      if (m < 0) continue;
      const int brow_i = p * RPP + trow;
      float v[VEC];
#pragma unroll
      for (int q = 0; q < VEC / 4; ++q) {
        const float4 x = *(const float4*)(ct + brow_i * CLD + tcol + 4 * q);
        v[4 * q] = x.x; v[4 * q + 1] = x.y; v[4 * q + 2] = x.z; v[4 * q + 3] = x.w;
      }
      const float brow = (g.bias_mode == 2) ? g.bias[m] : 0.f;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] += bv[e] + brow;
      TOUT* cp = C + (long)m * g.ldc + n;
      if (vec_ok && n + VEC <= g.N) {
        if constexpr (sizeof(TOUT) == 2) {
          if constexpr (RESID) if (R) {
            const unsigned int rw[4] = {rpre[i][p].x, rpre[i][p].y, rpre[i][p].z, rpre[i][p].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = res_apply(v[2 * e], bf2f(rw[e] & 0xffff), g.relu); v[2 * e + 1] = res_apply(v[2 * e + 1], bf2f(rw[e] >> 16), g.relu); }
          }
          if (g.relu == 1) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if constexpr (ABLATE == 3) { asm volatile("" ::"v"(v[0] + v[1] + v[2] + v[3] + v[4] + v[5] + v[6] + v[7])); }
          else *(uint4*)cp = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
          if constexpr (RESID) if (R) {
            v[0] = res_apply(v[0], __uint_as_float(rpre[i][p].x), g.relu); v[1] = res_apply(v[1], __uint_as_float(rpre[i][p].y), g.relu);
            v[2] = res_apply(v[2], __uint_as_float(rpre[i][p].z), g.relu); v[3] = res_apply(v[3], __uint_as_float(rpre[i][p].w), g.relu);
          }
          if (g.relu == 1) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *(float4*)cp = make_float4(v[0], v[1], v[2], v[3]);
        }
      } else {
        const TOUT* rp = R ? R + (long)m * g.ldc + n : nullptr;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (n + e < g.N) {
            float x = v[e];
            if (rp) x = res_apply(x, load_out<TOUT>(rp + e), g.relu);
            if (g.relu == 1) x = fmaxf(x, 0.f);
            store_out<TOUT>(cp + e, x);
          }
        }
      }
    }
  }
  }   // EPI
  }   // nt (row-panel loop)
  if (g.phase_ts && threadIdx.x == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    g.phase_ts[((long)blockIdx.x + (long)gridDim.x * blockIdx.y) * 8 + 3] = wall_clock64();
  }
}

// ---------------------------------------------------------------------------------------
// Row-panel kernel for the HBM-bound 1x1 "expand" convolutions / short-K GEMMs (K = 64 .. 512, N a multiple of 256):
// ONE workgroup owns BM output rows and ALL N columns.
//   * the A panel [BM][K] is loaded into LDS once and stays there (the tiled kernels re-stream it per column tile);
//   * W [N][K] streams through a 3-buffer ring of 32-deep slabs as ONE continuous sequence over (column tile, k-slab):
//     the slabs of the next column tile are already in flight while the current tile's epilogue runs, so the
//     load -> MFMA -> store phases of consecutive tiles overlap inside the workgroup (the 256x256 kernels run them
//     back to back at one workgroup per CU: 26 us per tile for 4 k-slabs, r02 ablation in DESIGN.md);
//   * the epilogue band (32 rows) has its own LDS, and the shortcut rows of the NEXT tile are fetched before the
//     current tile's stores are issued.
// vmcnt discipline: global stores and loads retire out of order with respect to each other, so the first wait of every
// column tile is vmcnt(0) (stores of the previous epilogue, shortcut prefetch, first slabs); inside a tile only
// LDS-direct loads are outstanding and the waits are counted (one slab stays in flight across the barrier).
// ---------------------------------------------------------------------------------------
template <int KP> __device__ __forceinline__ int panel_swz(int slot, int row) {
  if constexpr (KP / 8 >= 16) return (slot & ~15) | ((slot & 15) ^ (row & 15));
  else return slot ^ ((row >> 1) & 7);
}

template <int BM, int KP, bool RESID>
__global__ __launch_bounds__(512) void gemm_panel_kernel(GemmArgs g) {
  constexpr int WM = 2, WN = 4, NW = 8, NT = 512, BN = 256;
  constexpr int TM = BM / (32 * WM), TN = 2;
  constexpr int ROWA = KP * 2;                         // bytes per A row in LDS
  constexpr int A_BYTES = BM * ROWA;
  constexpr int A_INSTR = A_BYTES / 1024 / NW;         // LDS-direct loads per wave for the panel
  constexpr int BKW = 32, WROWB = 64, NSTAGE = 3, WSTAGE = BN * WROWB;      // 16 KiB slabs
  constexpr int WG_PER_WAVE = BN / (16 * NW);          // 16-row groups of a slab per wave = LDS-direct loads per slab
  constexpr int NKS = KP / BKW;                        // slabs per column tile
  constexpr int CLD = BN + 4, BROWS = 32;
  constexpr int BAND_BYTES = BROWS * CLD * 4;
  constexpr int LDS_BYTES = A_BYTES + NSTAGE * WSTAGE + BAND_BYTES;
  static_assert(A_INSTR >= 1 && LDS_BYTES <= 160 * 1024, "panel does not fit");
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  unsigned char* sA = lds;
  unsigned char* sW = lds + A_BYTES;
  float* ct = (float*)(lds + A_BYTES + NSTAGE * WSTAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave / WN, wc = wave % WN;
  const int m0 = blockIdx.x * BM;
  const unsigned short* A = (const unsigned short*)g.A;
  const unsigned short* W = (const unsigned short*)g.W;
  unsigned short* C = (unsigned short*)g.C;
  const unsigned short* R = (RESID && g.resid) ? (const unsigned short*)g.resid : nullptr;
  const int ntiles = g.N / BN;
  const int S = ntiles * NKS;                          // slabs of the whole W stream

  // ---- A panel: lane -> (row, 16-byte slot) of a 1 KiB piece, chunk permuted on the source side
#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int byte = ((wave * A_INSTR + j) << 10) + lane * 16;
    const int row = byte / ROWA, slot = (byte % ROWA) >> 4;
    const int gr = m0 + row;
    const void* src = (gr < g.M) ? (const void*)(A + (long)gr * g.lda + panel_swz<KP>(slot, row) * 8) : (const void*)g_zero16;
    __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(sA + ((wave * A_INSTR + j) << 10)), 16, 0, 0);
  }
  // ---- W ring: slab s = (column tile s / NKS, k-slab s % NKS); rows of 64 B, slot = chunk ^ ((row >> 2) & 3)
  const int wl_row = lane >> 2, wl_slot = lane & 3;
  auto stage_w = [&](int s) {
    const int tile = s / NKS, ks = s - tile * NKS;
    unsigned char* dst = sW + (s % NSTAGE) * WSTAGE + wave * (WG_PER_WAVE * 1024);
#pragma unroll
    for (int j = 0; j < WG_PER_WAVE; ++j) {
      const int tr_ = (wave * WG_PER_WAVE + j) * 16 + wl_row;
      const int gr = tile * BN + tr_;
      const void* src = (const void*)(W + (long)gr * g.ldw + ks * BKW + (wl_slot ^ ((tr_ >> 2) & 3)) * 8);
      __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(dst + j * 1024), 16, 0, 0);
    }
  };
  stage_w(0);
  if (S > 1) stage_w(1);

  // epilogue geometry: band of 32 rows x 256 columns, 32 threads per row (8 columns each), 16 rows per sweep
  constexpr int TPR = BN / 8, RPP = NT / TPR, NSW = BROWS / RPP;
  const int tcol = (tid % TPR) * 8, trow = tid / TPR;
  const bool vec_ok = ((g.ldc & 7) == 0) && ((((size_t)C) & 15) == 0) && (!R || (((size_t)R) & 15) == 0);
  auto out_row = [&](int pass, int sw) {               // pass = i * WM + wrow
    const int i = pass / WM, wrow = pass % WM;
    const int m = m0 + wrow * (BM / WM) + i * 32 + sw * RPP + trow;
    return m < g.M ? m : -1;
  };
  constexpr int NPASS = TM * WM;
  uint4 rcur[RESID ? NPASS : 1][RESID ? NSW : 1], rnext[RESID ? NPASS : 1][RESID ? NSW : 1];
  auto load_resid = [&](int tile, uint4 (&dst)[RESID ? NPASS : 1][RESID ? NSW : 1]) {
    if constexpr (RESID) {
#pragma unroll
      for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) {
          const int m = out_row(p, sw);
          dst[p][sw] = (R && vec_ok && m >= 0) ? *(const uint4*)(R + (long)m * g.ldc + tile * BN + tcol) : make_uint4(0, 0, 0, 0);
        }
    }
  };
  load_resid(0, rcur);

  for (int tile = 0; tile < ntiles; ++tile) {
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int ks = 0; ks < NKS; ++ks) {
      const int s = tile * NKS + ks;
      // first slab of a tile: drain everything (stores of the previous epilogue, shortcut rows, panel, first slabs);
      // afterwards only LDS-direct loads issued inside this loop are outstanding: leave the newest slab in flight
      if (ks == 0 || s + 1 >= S) wait_vm_barrier<0>();
      else wait_vm_barrier<WG_PER_WAVE>();
      if (s + 2 < S) stage_w(s + 2);
      const unsigned char* lb = sW + (s % NSTAGE) * WSTAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[TM], bfr[TN];
        const int ch = 2 * kk + (lane >> 5);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int row = wr * (BM / WM) + i * 32 + (lane & 31);
          af[i] = *(const bf16x8*)(sA + row * ROWA + (panel_swz<KP>(ks * 4 + ch, row) << 4));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int row = wc * 64 + j * 32 + (lane & 31);
          bfr[j] = *(const bf16x8*)(lb + row * WROWB + ((ch ^ ((row >> 2) & 3)) << 4));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
      }
    }
    // shortcut rows of the NEXT column tile: in flight while this tile's band passes run
    if (tile + 1 < ntiles) load_resid(tile + 1, rnext);

    const int n = tile * BN + tcol;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (g.bias_mode == 1) ? g.bias[n + e] : 0.f;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const int i = p / WM, wrow = p % WM;
      // (raw barriers: __syncthreads() would wait vmcnt(0) and expose the shortcut / slab prefetch issued above)
      if (p > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // previous band fully read
      if (wr == wrow) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int row = lane & 31;
            const int col = wc * 64 + j * 32 + 8 * gq + 4 * (lane >> 5);
            *(float4*)(ct + row * CLD + col) = make_float4(acc[i][j][4 * gq], acc[i][j][4 * gq + 1], acc[i][j][4 * gq + 2], acc[i][j][4 * gq + 3]);
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int sw = 0; sw < NSW; ++sw) {
        const int m = out_row(p, sw);
        if (m < 0) continue;
        const int brow = sw * RPP + trow;
        float v[8];
        const float4 x0 = *(const float4*)(ct + brow * CLD + tcol), x1 = *(const float4*)(ct + brow * CLD + tcol + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        unsigned short* cp = C + (long)m * g.ldc + n;
        if (vec_ok) {
          if constexpr (RESID) if (R) {
            const unsigned int rw[4] = {rcur[p][sw].x, rcur[p][sw].y, rcur[p][sw].z, rcur[p][sw].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = res_apply(v[2 * e], bf2f(rw[e] & 0xffff), g.relu); v[2 * e + 1] = res_apply(v[2 * e + 1], bf2f(rw[e] >> 16), g.relu); }
          }
          if (g.relu == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *(uint4*)cp = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
          const unsigned short* rp = R ? R + (long)m * g.ldc + n : nullptr;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e];
            if (rp) x = res_apply(x, bf2f(rp[e]), g.relu);
            if (g.relu == 1) x = fmaxf(x, 0.f);
            cp[e] = f2bf(x);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // band free for the next tile
    if constexpr (RESID) {
#pragma unroll
      for (int p = 0; p < NPASS; ++p)
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) rcur[p][sw] = rnext[p][sw];
    }
  }
}

// ---------------------------------------------------------------------------------------
// Row-panel kernel, second form: the A panel [BM][K] is LDS resident as above, but W never touches LDS.  It is
// pre-packed ONCE (model load) in MFMA fragment order -- block (column tile of 32, k-step of 16) = 64 lanes x 16 bytes,
// lane l = W[n0 + (l & 31)][k0 + 8 (l >> 5) .. +8] -- so that every fragment load is one contiguous 1 KiB request, and
// each wave streams the fragments of ITS 32 columns through a register ring D k-steps deep.  Consequences:
//   * no barrier in the main loop (A is static, W is private to the wave): waves drift apart and hide each other's waits;
//   * 8 waves x D loads x 1 KiB = 64 KiB of W in flight per CU on top of the shortcut rows and stores, without spending
//     LDS on it -- the LDS-ring form above can keep 32 KiB in flight and is latency bound at ~0.1 us of MFMA per slab;
//   * the ring keeps running across column tiles: the first fragments of the next tile are requested during the last
//     k-steps of the current one and land while its epilogue runs.
// Wave grid 1 x 8: wave w owns columns [256 tile + 32 w, +32) of all BM rows (TM = BM / 32 accumulator tiles).
// ---------------------------------------------------------------------------------------
// OCC = 2: two workgroups per CU (BM = 64: 32 KiB panel + 33 KiB band each, <= 128 VGPRs): while one workgroup waits for
// its stores to be acknowledged (loads and stores share vmcnt, so the first fragment use after an epilogue drains both)
// the other one computes.  The shortcut rows are then fetched at the start of their own tile (no second register set).
template <int BM, int KP, bool RESID, int OCC = 1>
__global__ __launch_bounds__(512, 2 * OCC) void gemm_panelw_kernel(GemmArgs g) {
  constexpr int NT = 512, BN = 256;
  constexpr int TM = BM / 32;
  constexpr int ROWA = KP * 2, A_BYTES = BM * ROWA, A_INSTR = A_BYTES / 1024 / 8;
  constexpr int NK = KP / 16;                          // k-steps per column tile
  constexpr int D = NK < 8 ? NK : 8;                   // fragment ring depth
  static_assert(NK % D == 0 && A_INSTR >= 1, "ring depth must divide the k-steps of a tile");
  constexpr int CLD = BN + 4, BROWS = 32;
  constexpr int LDS_BYTES = A_BYTES + BROWS * CLD * 4;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
  unsigned char* sA = lds;
  float* ct = (float*)(lds + A_BYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = blockIdx.x * BM;
  const unsigned short* A = (const unsigned short*)g.A;
  const uint4* Wf = (const uint4*)g.Wf;
  unsigned short* C = (unsigned short*)g.C;
  const unsigned short* R = (RESID && g.resid) ? (const unsigned short*)g.resid : nullptr;
  const int ntiles = g.N / BN;

#pragma unroll
  for (int j = 0; j < A_INSTR; ++j) {
    const int byte = ((wave * A_INSTR + j) << 10) + lane * 16;
    const int row = byte / ROWA, slot = (byte % ROWA) >> 4;
    const int gr = m0 + row;
    const void* src = (gr < g.M) ? (const void*)(A + (long)gr * g.lda + panel_swz<KP>(slot, row) * 8) : (const void*)g_zero16;
    __builtin_amdgcn_global_load_lds((gas_ptr)src, (las_ptr)(sA + ((wave * A_INSTR + j) << 10)), 16, 0, 0);
  }
  // fragment stream of this wave: column tile of 32 = tile * 8 + wave; blocks of 64 uint4 per k-step
  auto frag_ptr = [&](int tile, int ks) { return Wf + ((long)(tile * 8 + wave) * NK + ks) * 64 + lane; };
  uint4 ring[D];
#pragma unroll
  for (int d = 0; d < D; ++d) ring[d] = *frag_ptr(0, d);

  constexpr int TPR = BN / 8, RPP = NT / TPR, NSW = BROWS / RPP;
  const int tcol = (tid % TPR) * 8, trow = tid / TPR;
  const bool vec_ok = ((g.ldc & 7) == 0) && ((((size_t)C) & 15) == 0) && (!R || (((size_t)R) & 15) == 0);
  auto out_row = [&](int i, int sw) {
    const int m = m0 + i * 32 + sw * RPP + trow;
    return m < g.M ? m : -1;
  };
  constexpr bool PREF = RESID && OCC == 1;             // shortcut rows of tile t+1 requested before the epilogue of tile t
  uint4 rcur[RESID ? TM : 1][RESID ? NSW : 1], rnext[PREF ? TM : 1][PREF ? NSW : 1];
  auto load_resid = [&](int tile, auto& dst) {
    if constexpr (RESID) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) {
          const int m = out_row(i, sw);
          dst[i][sw] = (R && vec_ok && m >= 0) ? *(const uint4*)(R + (long)m * g.ldc + tile * BN + tcol) : make_uint4(0, 0, 0, 0);
        }
    }
  };
  if constexpr (PREF) load_resid(0, rcur);
  __syncthreads();                                     // A panel landed (drains the LDS-direct loads)

  for (int tile = 0; tile < ntiles; ++tile) {
    f32x16 acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const int tnext = tile + 1 < ntiles ? tile + 1 : tile;           // (last tile: harmless re-read of its own fragments)
    if constexpr (RESID && !PREF) load_resid(tile, rcur);
#pragma unroll
    for (int ks = 0; ks < NK; ++ks) {
      bf16x8 af[TM];
      const int ch = 2 * ks + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(sA + row * ROWA + (panel_swz<KP>(ch, row) << 4));
      }
      bf16x8 wfr;
      *(uint4*)&wfr = ring[ks % D];
#pragma unroll
      for (int i = 0; i < TM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfr, af[i], acc[i], 0, 0, 0);
      // refill this ring slot with the fragment D k-steps ahead (possibly of the next column tile)
      ring[ks % D] = (ks + D < NK) ? *frag_ptr(tile, ks + D) : *frag_ptr(tnext, ks + D - NK);
    }
    if constexpr (PREF) { if (tile + 1 < ntiles) load_resid(tile + 1, rnext); }

    const int n = tile * BN + tcol;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (g.bias_mode == 1) ? g.bias[n + e] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (i > 0) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // previous band fully read
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int row = lane & 31;
        const int col = wave * 32 + 8 * gq + 4 * (lane >> 5);
        *(float4*)(ct + row * CLD + col) = make_float4(acc[i][4 * gq], acc[i][4 * gq + 1], acc[i][4 * gq + 2], acc[i][4 * gq + 3]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int sw = 0; sw < NSW; ++sw) {
        const int m = out_row(i, sw);
        if (m < 0) continue;
        const int brow = sw * RPP + trow;
        float v[8];
        const float4 x0 = *(const float4*)(ct + brow * CLD + tcol), x1 = *(const float4*)(ct + brow * CLD + tcol + 4);
        v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += bv[e];
        unsigned short* cp = C + (long)m * g.ldc + n;
        if (vec_ok) {
          if constexpr (RESID) if (R) {
            const unsigned int rw[4] = {rcur[i][sw].x, rcur[i][sw].y, rcur[i][sw].z, rcur[i][sw].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] = res_apply(v[2 * e], bf2f(rw[e] & 0xffff), g.relu); v[2 * e + 1] = res_apply(v[2 * e + 1], bf2f(rw[e] >> 16), g.relu); }
          }
          if (g.relu == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          *(uint4*)cp = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
        } else {
          const unsigned short* rp = R ? R + (long)m * g.ldc + n : nullptr;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float x = v[e];
            if (rp) x = res_apply(x, bf2f(rp[e]), g.relu);
            if (g.relu == 1) x = fmaxf(x, 0.f);
            cp[e] = f2bf(x);
          }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                   // band free for the next tile
    if constexpr (PREF) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int sw = 0; sw < NSW; ++sw) rcur[i][sw] = rnext[i][sw];
    }
  }
}

// W [N][K] (row stride ldw) -> fragment order: block (n / 32, k / 16) = 64 x 16 bytes, lane l = W[32 nb + (l & 31)][16 kb + 8 (l >> 5) ..]
__global__ __launch_bounds__(256) void pack_w_frag_kernel(const unsigned short* w, long ldw, uint4* out, int N, int K) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)(N / 32) * (K / 16) * 64;
  if (t >= total) return;
  const int l = (int)(t & 63);
  const long blk = t >> 6;
  const int kb = (int)(blk % (K / 16)), nb = (int)(blk / (K / 16));
  out[t] = *(const uint4*)(w + (long)(nb * 32 + (l & 31)) * ldw + kb * 16 + 8 * (l >> 5));
}

// ---------------------------------------------------------------------------------------
// f32 in / f32 accumulate (exact fp32 MFMA, bit-wise an fmaf chain).  128x128 tile.
// ---------------------------------------------------------------------------------------
// CONV: the A operand is an NHWC float32 image batch read as an implicit im2col matrix (GemmArgs c* fields, Cin % 16 == 0 so that a
// 16-deep k block lies inside one filter tap): row m = output pixel (b, oy, ox), k = (r S + s) Cin + ic; a padded tap contributes
// exact zeros.  The float32 parity path of every convolution of the graph (relnet_conv2d_nhwc_f32): same fmaf chain per output
// element as the plain GEMM, whatever the tap order -- k is walked in the order of the packed weight row.
template <typename TOUT, bool CONV = false>
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(GemmArgs g) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * 128 + wr * 64, n0 = blockIdx.x * 128 + wc * 64;
  const float* A = (const float*)g.A + (long)blockIdx.z * g.strideA;
  const float* W = (const float*)g.W + (long)blockIdx.z * g.strideW;
  TOUT* C = (TOUT*)g.C + (long)blockIdx.z * g.strideC;
  const TOUT* R = g.resid ? (const TOUT*)g.resid + (long)blockIdx.z * g.strideC : nullptr;
  const int half = lane >> 5;
  // rows beyond M/N are clamped for the loads; their products are never stored
  const float* pa[2];
  const float* pb[2];
  int iy0[2], ix0[2];            // CONV: input coordinates of tap (0, 0) of this lane's two output pixels
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = m0 + i * 32 + (lane & 31); ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + i * 32 + (lane & 31); rb = rb < g.N ? rb : g.N - 1;
    if constexpr (CONV) {
      const int hw = g.cHout * g.cWout;
      const int b = ra / hw, rem = ra - b * hw, oy = rem / g.cWout, ox = rem - oy * g.cWout;
      iy0[i] = oy * g.cStride - g.cPad; ix0[i] = ox * g.cStride - g.cPad;
      pa[i] = A + (long)b * g.cImg + half * 8;
    } else {
      pa[i] = A + (long)ra * g.lda + half * 8;
    }
    pb[i] = W + (long)rb * g.ldw + half * 8;
  }
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 a[2][2], b[2][2], an[2][2], bn[2][2];
  int tr = 0, ts = 0, tc = 0;      // CONV: filter tap (row, column) and first channel of the k block the NEXT load fetches (loads are issued in k order)
  auto load = [&](int kb, float4 (&x)[2][2], float4 (&y)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if constexpr (CONV) {
        const int iy = iy0[i] + tr * g.cDil, ix = ix0[i] + ts * g.cDil;
        if (iy >= 0 && iy < g.cH && ix >= 0 && ix < g.cW) {
          const float* p = pa[i] + ((long)iy * g.cW + ix) * g.cPix + tc;
          x[i][0] = *(const float4*)p;
          x[i][1] = *(const float4*)(p + 4);
        } else {
          x[i][0] = make_float4(0.f, 0.f, 0.f, 0.f);
          x[i][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      } else {
        x[i][0] = *(const float4*)(pa[i] + kb * 16);
        x[i][1] = *(const float4*)(pa[i] + kb * 16 + 4);
      }
      y[i][0] = *(const float4*)(pb[i] + kb * 16);
      y[i][1] = *(const float4*)(pb[i] + kb * 16 + 4);
    }
    if constexpr (CONV) {
      tc += 16;
      if (tc == g.cCin) { tc = 0; if (++ts == g.cS) { ts = 0; ++tr; } }
    }
  };
  const int nkb = g.K / 16;
  load(0, a, b);
  for (int kb = 0; kb < nkb; ++kb) {
    if (kb + 1 < nkb) load(kb + 1, an, bn);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float av[2], bv[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float4& xa = a[i][s >> 2];
        const float4& xb = b[i][s >> 2];
        av[i] = (s & 3) == 0 ? xa.x : (s & 3) == 1 ? xa.y : (s & 3) == 2 ? xa.z : xa.w;
        bv[i] = (s & 3) == 0 ? xb.x : (s & 3) == 1 ? xb.y : (s & 3) == 2 ? xb.z : xb.w;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kb + 1 < nkb) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 2; ++q) { a[i][q] = an[i][q]; b[i][q] = bn[i][q]; }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      epilogue_tile<TOUT>(g, C, R, acc[i][j], m0 + i * 32, n0 + j * 32, lane);
}

}  // namespace relnet

using namespace relnet;

// dtype codes shared by the whole C-ABI
enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

static int g_ablate = 0;         // measurement knob (ring tiles, conv mode, bf16 out, no shortcut): 1 = fill path only, 2 = LDS + MFMA only
extern "C" void relnet_gemm_debug_ablate(int a) { g_ablate = a; }
static int g_force_tile = 0;     // tuning knob: 0 auto, else index into the config list below
static int g_force_nloop = 0;    // tuning knob: 0 auto, else column tiles per workgroup
static int g_swizzle = 1;        // tuning knob: XCD-aware tile order (0 = plain blockIdx order)
extern "C" void relnet_gemm_force_tile(int t) { g_force_tile = t; }
extern "C" int relnet_gemm_get_forced_tile(void) { return g_force_tile; }
extern "C" void relnet_gemm_force_nloop(int n) { g_force_nloop = n; }
extern "C" void relnet_gemm_set_swizzle(int on) { g_swizzle = on; }
static int g_korder = 1;         // tuning knob: 1 = (channel chunk, tap) k order + XCD-contiguous row tiles for R*S > 1 ring launches
extern "C" void relnet_gemm_debug_korder(int on) { g_korder = on; }
static long long* g_phase_ts = nullptr;   // measurement knob: ring kernels write their phase timestamps there (8 words per workgroup)
extern "C" void relnet_gemm_debug_phase_ts(void* buf) { g_phase_ts = (long long*)buf; }
static int g_asm = 1;            // tuning knob: 0 = pick_tile never chooses tiles 18 / 19 (hand-scheduled k-loops)
extern "C" void relnet_gemm_debug_asm(int on) { g_asm = on; }

// ---- split-K work area (relnet_gemm_set_workspace): 16 KB of tile counters (zero between launches) + fp32 partial tiles.  The area is the CALLER's:
// the launches that share one must be ordered (one stream, or the dependencies of one captured graph), so the host side (ops.py) keeps one area per
// launching stream and names the current one before a GEMM call; the pointer is per host thread.  Nothing is allocated here (capture-safe).
constexpr long kSkCounterBytes = 16384;
struct SplitKArea { unsigned char* base = nullptr; long bytes = 0; };
static thread_local SplitKArea t_sk;
static int g_splitk = 0;         // tuning knob: 0 auto, 1 = never split, k >= 2 = split by k wherever the split-K tile is chosen, -2 = auto incl. the 129..320-tile launches
extern "C" void relnet_gemm_debug_splitk(int k) { g_splitk = k; }
extern "C" int relnet_gemm_set_workspace(void* ws, long bytes) {
  RELNET_REQUIRE(ws == nullptr || (bytes >= kSkCounterBytes + (1L << 20) && (((uintptr_t)ws) & 255) == 0),
                 "relnet_gemm_set_workspace: need a 256-byte aligned area of at least %ld bytes", kSkCounterBytes + (1L << 20));
  t_sk.base = (unsigned char*)ws; t_sk.bytes = ws ? bytes : 0;
  return 0;
}
// room for `need` bytes of partial tiles and `tiles` counters in the current work area?
static bool splitk_area(long need, long tiles, float** part, unsigned int** cnt) {
  if (!t_sk.base || tiles * 4 > kSkCounterBytes || need > t_sk.bytes - kSkCounterBytes) return false;
  *cnt = (unsigned int*)t_sk.base;
  *part = (float*)(t_sk.base + kSkCounterBytes);
  return true;
}
// how many ways the k-loop of a 64 x 64-tiled launch is split: only launches of at most one workgroup per CU (every k-step of such a launch is an
// exposed L2 round trip, so the k-loop length IS the launch time), enough to put ~2 workgroups on every CU, at least 4 k-slabs per workgroup
static int splitk_ways(long M, long N, long K, int batch, const GemmArgs& g) {
  if (g_splitk == 1 || batch != 1 || (N & 7) || (g.ldc & 7) || g.mask || ((((uintptr_t)g.C) | ((uintptr_t)g.resid)) & 15)) return 1;
  const long tiles = ((M + 63) / 64) * ((N + 63) / 64), nk = K / 64;
  if (g_splitk >= 2) return (int)(nk / g_splitk >= 1 ? (g_splitk > 8 ? 8 : g_splitk) : 1);
  // Measured on the one-image shapes (tools/splitk_probe.py, profiles/r06_notes/splitk_one_image.txt): the partial tiles cross XCDs, i.e. they travel
  // through memory (sc1 stores, L2-bypassing loads), which costs the last arriver ~8 us -- as much as 4 - 5 k-steps.  In isolation the split pays only
  // for the long k-loops: fc_new_1 (196 k-slabs, 80 tiles: 60.8 -> 34.7 us at 4 ways) and rpn_conv_3x3 (144 k-slabs, 304 tiles: 101 -> 78 us at 3 ways);
  // res4 3x3 (36 k-slabs) stays at 21 us either way, res5 3x3 (72) at 52 us.  Inside the one-image step (same-box A/B, tools/scripts/r06_ab.sh,
  // 2 000 replays each): unsplit 2.128 / 2.133 / 2.139 ms, fc_new_1 only 2.135 / 2.139 / 2.131 ms, both 2.192 / 2.202 / 2.167 ms -- the 912 workgroups
  // of a split rpn_conv_3x3 on the side stream take CUs from res5 on the main one.  Default: the <= 128-tile launches only (neutral for inference,
  // -0.04 ms on the one-image training step); g_splitk = -2 adds the <= 320-tile ones.
  if (tiles > 320 || nk < 128) return 1;
  if (g_splitk != -2 && tiles > 128) return 1;
  return tiles <= 128 ? 4 : 3;
}

template <int BM, int BN, int WM, int WN, int CONV, int KU = 1>
static void launch_cfg(GemmArgs g, int batch, int out_dtype, hipStream_t s, int ksplit = 1) {
  const int ntile = (g.N + BN - 1) / BN;
  if (ksplit > 1) {
    const long tiles = (long)ntile * ((g.M + BM - 1) / BM);
    if (batch == 1 && splitk_area((long)ksplit * g.M * g.N * 4, tiles, &g.kpart, &g.kcnt)) {
      g.ksplit = ksplit; g.n_loop = 1;
      dim3 grid(ntile, (g.M + BM - 1) / BM, ksplit);
      g.xcd_swizzle = (g_swizzle && ntile > 1 && tiles >= 16) ? 1 : 0;
      if (out_dtype == RELNET_BF16) gemm_nt_bf16_kernel<BM, BN, WM, WN, unsigned short, CONV, KU><<<grid, 64 * WM * WN, 0, s>>>(g);
      else gemm_nt_bf16_kernel<BM, BN, WM, WN, float, CONV, KU><<<grid, 64 * WM * WN, 0, s>>>(g);
      return;
    }
  }
  int nloop = g_force_nloop > 0 ? g_force_nloop : g.n_loop;
  const bool swz = g_swizzle && batch == 1 && ntile > 1 && (long)ntile * ((g.M + BM - 1) / BM) >= 16;
  // with the XCD-aware order the column tiles of a row panel run side by side on one XCD and share the A rows through
  // its L2; walking them serially in one workgroup (row-panel mode) only costs parallelism then (measured, 54 images:
  // res4 expand 229.9 us plain/auto -> 217.7 us swizzle/n_loop 1; per-step convolution total 23.07 -> 22.36 ms)
  if (swz && g_force_nloop == 0) nloop = 1;
  if (nloop < 1) nloop = 1;
  if (nloop > ntile) nloop = ntile;
  g.n_loop = nloop;
  dim3 grid((ntile + nloop - 1) / nloop, (g.M + BM - 1) / BM, batch);
  g.xcd_swizzle = (swz && grid.x > 1) ? 1 : 0;
  if (out_dtype == RELNET_BF16) gemm_nt_bf16_kernel<BM, BN, WM, WN, unsigned short, CONV, KU><<<grid, 64 * WM * WN, 0, s>>>(g);
  else gemm_nt_bf16_kernel<BM, BN, WM, WN, float, CONV, KU><<<grid, 64 * WM * WN, 0, s>>>(g);
}

template <int BM, int BN, int WM, int WN, int CONV, int BK, int NSTAGE, int SCHED = 0, int EPI = 0>
static void launch_ring(GemmArgs g, int batch, int out_dtype, hipStream_t s) {
  const int ntile = (g.N + BN - 1) / BN;
  int nloop = g_force_nloop > 0 ? g_force_nloop : g.n_loop;
  bool swz = g_swizzle && batch == 1 && ntile > 1 && (long)ntile * ((g.M + BM - 1) / BM) >= 16;
  // ... unless the filters alone fill an XCD's L2 (res5a's projection, 1024 -> 2048: 4 MB): the XCD-aware order then re-streams them per
  // row panel; in plain order column tile c runs on XCD c % 8 and its 512 KB filter tile stays put while the row panels stream past
  // (tools/swizzle_probe.py, 54 images: 727 -> 645 us; the two-column-tile layers are indifferent)
  if (g_swizzle == 1 && ntile >= 8 && (long)g.N * g.K * 2 >= (4L << 20)) swz = false;
  if ((swz && g_force_nloop == 0) || SCHED == 2) nloop = 1;
  if (nloop < 1) nloop = 1;
  if (nloop > ntile) nloop = ntile;
  g.n_loop = nloop;
  dim3 grid((ntile + nloop - 1) / nloop, (g.M + BM - 1) / BM, batch);
  g.xcd_swizzle = (swz && grid.x > 1) ? 1 : 0;
  g.korder = 0;
  g.phase_ts = g_phase_ts;
  if constexpr (CONV == 1) {
    // Spatial convolutions re-read every input pixel once per tap.  In (tap, channel chunk) order the re-reads of one workgroup are
    // R * S k-slabs apart and the 32 resident workgroups of an XCD stream ~24 MB in between: the 4 MiB L2 has long dropped the
    // lines and every tap is served by the Infinity Cache (the ~10 TB/s ceiling of the fill-only ablation).  In (channel chunk, tap)
    // order the nine taps of a 64-channel chunk touch the same ~40 KB per workgroup back to back (1.3 MB per XCD: L2 hits), and with
    // consecutive row tiles on one XCD neighbouring tiles share their halo rows in that L2 as well.  Same products, different fp32
    // summation order.
    if (g_korder && g.cR * g.cS > 1 && g.cCin % BK == 0) {
      g.korder = 1;
      if (g_swizzle && batch == 1 && grid.x == 1 && grid.y >= 16) g.xcd_swizzle = 1;
    }
  }
  const int nthr = 64 * WM * WN;
  if constexpr (CONV == 1) {
    if (g_ablate && g_ablate <= 2 && out_dtype == RELNET_BF16 && !g.resid) {
      constexpr int ASCHED = (SCHED == 5 || SCHED == 6) ? 0 : SCHED;          // (the asm k-loops have no ablation forms)
      if (g_ablate == 1) gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, false, ASCHED, 1><<<grid, nthr, 0, s>>>(g);
      else gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, false, ASCHED, 2><<<grid, nthr, 0, s>>>(g);
      return;
    }
    if (g_ablate == 3 && out_dtype == RELNET_BF16) {           // no output stores
      if (g.resid) gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, true, 0, 3><<<grid, nthr, 0, s>>>(g);
      else gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, false, 0, 3><<<grid, nthr, 0, s>>>(g);
      return;
    }
  }
  if constexpr (EPI == 1) {
    if (g.resid) {
      if (out_dtype == RELNET_BF16) gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, true, 0, 0, 1><<<grid, nthr, 0, s>>>(g);
      else gemm_ring_kernel<BM, BN, WM, WN, float, CONV, BK, NSTAGE, true, 0, 0, 1><<<grid, nthr, 0, s>>>(g);
    } else {
      if (out_dtype == RELNET_BF16) gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, false, SCHED, 0, 1><<<grid, nthr, 0, s>>>(g);
      else gemm_ring_kernel<BM, BN, WM, WN, float, CONV, BK, NSTAGE, false, SCHED, 0, 1><<<grid, nthr, 0, s>>>(g);
    }
    return;
  }
  if (g.resid) {
    if (out_dtype == RELNET_BF16) gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, true><<<grid, nthr, 0, s>>>(g);
    else gemm_ring_kernel<BM, BN, WM, WN, float, CONV, BK, NSTAGE, true><<<grid, nthr, 0, s>>>(g);
  } else {
    // (the fragment-pipelined schedule needs the registers the residual prefetch would take: shortcut-free layers only)
    if (out_dtype == RELNET_BF16) gemm_ring_kernel<BM, BN, WM, WN, unsigned short, CONV, BK, NSTAGE, false, SCHED><<<grid, nthr, 0, s>>>(g);
    else gemm_ring_kernel<BM, BN, WM, WN, float, CONV, BK, NSTAGE, false, SCHED><<<grid, nthr, 0, s>>>(g);
  }
}

// configs: 1 = 256x256 (8 waves) 2 = 256x128 (8 waves) 3 = 128x128 (4 waves) 4 = 128x64 5 = 64x64   (2 LDS stages, BK 64)
// deep-pipeline (gemm_ring_kernel): 6 = 256x256, BK 32, 4 buffers   7 = 256x128, BK 64, 3 buffers
//                                   8 = 256x256, BK 64, 2 buffers (conflict-free swizzle only: the A/B of that change)
//                                   9 = 6 and 10 = 7 with the fragment-pipelined schedule on shortcut-free layers
//                                   11 = 8 with the register epilogue (no LDS band)
//                                   12 = 8 with the fragment-pipelined schedule on shortcut-free layers
// row-panel (gemm_panel_kernel):    13 = A panel resident, all column tiles per workgroup (K in {64,128,256,512}, N % 256 == 0,
//                                        bf16 out, 1x1 / plain GEMM); other shapes under 13 run configuration 1
//                                   14 = the same with W streamed from its fragment-order copy through registers
//                                        (gemm_panelw_kernel; needs the Wf operand, else configuration 13)
//                                   15 = 14 with 64-row panels, two workgroups per CU
// (measured and dropped: tile 8 with FOUR wavefronts of 128x128 -- 4x4 fragments, accumulators in AGPRs, half the LDS fragment
//  reads per MFMA -- is 20-40 % slower on every layer shape: one wavefront per SIMD cannot cover the LDS / MFMA latencies)
//                                   16 = 8 with the ping-pong (two wave rows half a phase apart) schedule on shortcut-free layers
// (measured and dropped, r03: 64x64 / 128x64 ring tiles with 3 / 2 k-slabs in flight for the small-batch steps: 20-50 % slower than
//  tiles 5 / 4 at 1 and 8 images -- four to five resident 2-stage workgroups per CU already cover the load latency)
//                                   17 = 8 with the A operand of 3x3 / stride-1 layers staged once per channel chunk as a pixel WINDOW shared by
//                                        the nine taps (other shapes under 17 run configuration 8)
//                                   18 = 8 with the hand-scheduled (inline asm) k-loop on shortcut-free layers whose N is a multiple of 256
//                                   19 = 18 with the asymmetric ring (activations three slots / two slabs ahead, filters two slots)
//                                   20 = 5 with two k-slabs per barrier (launches of less than one workgroup per CU: the 1-image step)
//                                   21 = 5 with four
//                                   22 = 192x128 (see launch_bf16)
//                                   23 = 20 with the k-loop SPLIT over several workgroups per tile (round 6; needs relnet_gemm_set_workspace, else = 20)
enum { GEMM_TILE_COUNT = 23 };

template <int CONV>
static bool launch_panel(const GemmArgs& g0, int batch, int out_dtype, hipStream_t s, bool use_wf = false, bool occ2 = false) {
  GemmArgs g = g0;
  if (batch != 1 || out_dtype != RELNET_BF16 || g.N % 256 != 0 || (g.ldc & 7)) return false;
  if constexpr (CONV == 2) return false;
  if constexpr (CONV == 1) {                             // 1x1, stride 1, no padding: a plain GEMM over the pixels
    if (g.cR != 1 || g.cS != 1 || g.cStride != 1 || g.cPad != 0) return false;
    if (g.cImg != (long)g.cH * g.cW * g.cPix) return false;          // images must be contiguous pixel runs
    g.lda = g.cPix;
  }
  const bool res = g.resid != nullptr;
  if (use_wf && g.Wf && occ2) {
#define RELNET_PANELW2(KP_)                                                                            \
  do {                                                                                                  \
    dim3 grid((g.M + 63) / 64);                                                                         \
    if (res) gemm_panelw_kernel<64, KP_, true, 2><<<grid, 512, 0, s>>>(g);                              \
    else gemm_panelw_kernel<64, KP_, false, 2><<<grid, 512, 0, s>>>(g);                                 \
    return true;                                                                                        \
  } while (0)
    switch (g.K) {
      case 64: RELNET_PANELW2(64);
      case 128: RELNET_PANELW2(128);
      case 256: RELNET_PANELW2(256);
      default: break;                              // (K = 512: a 64-row panel + band is 97 KiB, one workgroup per CU anyway)
    }
#undef RELNET_PANELW2
  }
  if (use_wf && g.Wf) {
#define RELNET_PANELW(BM_, KP_)                                                                        \
  do {                                                                                                  \
    dim3 grid((g.M + BM_ - 1) / BM_);                                                                   \
    if (res) gemm_panelw_kernel<BM_, KP_, true><<<grid, 512, 0, s>>>(g);                                \
    else gemm_panelw_kernel<BM_, KP_, false><<<grid, 512, 0, s>>>(g);                                   \
    return true;                                                                                        \
  } while (0)
    switch (g.K) {
      case 64: RELNET_PANELW(128, 64);
      case 128: RELNET_PANELW(128, 128);
      case 256: RELNET_PANELW(128, 256);
      case 512: RELNET_PANELW(64, 512);
      default: break;
    }
#undef RELNET_PANELW
  }
#define RELNET_PANEL(BM_, KP_)                                                                         \
  do {                                                                                                  \
    dim3 grid((g.M + BM_ - 1) / BM_);                                                                   \
    if (res) gemm_panel_kernel<BM_, KP_, true><<<grid, 512, 0, s>>>(g);                                 \
    else gemm_panel_kernel<BM_, KP_, false><<<grid, 512, 0, s>>>(g);                                    \
    return true;                                                                                        \
  } while (0)
  switch (g.K) {
    case 64: RELNET_PANEL(128, 64);
    case 128: RELNET_PANEL(128, 128);
    case 256: RELNET_PANEL(128, 256);
    case 512: RELNET_PANEL(64, 512);
    default: return false;
  }
#undef RELNET_PANEL
}
static int pick_tile(long M, long N, long K, int batch, int out_dtype, int has_resid, int has_wf = 0) {
  // Picked from on-device timings of every GEMM / convolution shape of the detector at 16
  // images per launch (tools/bench_gemm.py; table in DESIGN.md): wide tiles cut the
  // L2 -> LDS fill traffic of the compute-bound 3x3 / large-K layers, 128-row tiles keep more
  // workgroups resident for the short-K, store-bound expand convolutions.
  int cfg;
  if (N <= 64) cfg = 4;
  else if (M * batch <= 8192) cfg = (N >= 256 ? 4 : 5);
  else if (N <= 128) cfg = 3;
  else if (K <= 128) cfg = 4;
  else if (N % 256 != 0) cfg = 3;
  else cfg = 1;
  // fp32 outputs (weight gradients, column gradients): the 256x256 epilogue needs a second fp32 staging tile and
  // spills ~180 VGPRs; 256x128 holds everything in registers
  if (cfg == 1 && out_dtype != RELNET_BF16) cfg = 2;
  // shortcut-free 256x256 layers (3x3 / reduce convolutions, FC layers): the ring kernel's conflict-free LDS image and
  // residual-free register budget are worth 3-11 % (r02 tile table: res5 3x3 642 -> 574 us, rpn 3x3 1244 -> 1131 us)
  if (cfg == 1 && !has_resid) cfg = 8;
  // ... and from K = 512 the hand-scheduled k-loop with the asymmetric ring (tile 19; tile 18 = the same loop on the symmetric
  // two-slab ring).  r04, 54 images, us per launch, tile 8 / 18 / 19: res4 3x3 165 / 148 / 143, res5 3x3 603 / 515 / 503,
  // rpn 3x3 1157 / 972 / 950, res5 reduce 327 / 299 / 281, res4 reduce (K = 1024, HBM-fed) 97 / 99 / 90, conv_new_1 180 / 170 / 154
  if (cfg == 8 && K >= 512 && N % 256 == 0 && g_asm) cfg = 19;
  // res4 expand convolutions (256 -> 1024 + shortcut, 23 per step): with the weights also available in fragment order the
  // panel kernel reads A once and never stages W in LDS: 219 -> 191..203 us (r02 tile table; every other shape is slower)
  if (cfg == 1 && has_resid && has_wf && K == 256 && batch == 1) cfg = 14;
  // small problems (1 - 8 images per launch: the reference's BATCH_IMAGES = 1 protocol, the training step): a 256 x 256 grid
  // of < 128 workgroups leaves half of the 256 CUs idle.  Measured per shape with tools/bench_tiles.py (r03): at 8 images
  // res4 3x3 / reduce 73.5 / 38.9 us (75 workgroups of 256 x 256) -> 50.7 / 26.4 us on 128 x 64 tiles (600 workgroups);
  // at 1 image 64 x 64 tiles win on every N >= 256 layer (res4 3x3 31.3 -> 21.6 us, rpn 3x3 120 -> 101 us): per-step
  // convolution totals 5.60 -> 4.6 ms (8 images), 2.14 -> 1.7 ms (1 image).
  auto wgs = [&](long bm, long bn) { return ((M + bm - 1) / bm) * ((N + bn - 1) / bn) * (long)batch; };
  const int small = wgs(128, 64) >= 200 ? 4 : 5;
  if ((cfg == 1 || cfg == 8 || cfg == 18 || cfg == 19) && wgs(256, 256) < 128) cfg = small;
  // shortcut / mask layers (HBM-bound, K <= 512: four to eight k-slabs) on fewer than three rounds of 256 x 256 workgroups -- the training step's
  // 19 152-pixel maps: 300 workgroups = 1.17 rounds -- run better on 128 x 64 tiles whose 2 400 workgroups keep every CU's load queue full.
  // r05, 8 images, us per launch, tile 1 / tile 4 (profiles/r05_notes/tiles_b8_train_shapes.txt): res4 reduce-dgrad + shortcut 48.9 / 36.7,
  // res4 expand + shortcut 50.2 / 38.1, res5 expand 98.3 / 86.0, res5 reduce-dgrad 98.0 / 85.9
  else if (cfg == 1 && has_resid && K <= 512 && wgs(256, 256) < 768) cfg = 4;
  else if (cfg == 2 && wgs(256, 128) < 128) cfg = small;
  else if (cfg == 3 && wgs(128, 128) < 200) cfg = small;
  else if (cfg == 3 && N == 128 && K >= 512 && wgs(192, 128) >= 200) cfg = 22;       // res3 3x3 (K = 1152) and reduce / expand-dgrad (K = 512)
  // masked data gradients of res5 at the training step's size (N = 512, K = 2048 / 4608, 150 workgroups of 256 x 256): 142 -> 129 us, 73 -> 66 us
  else if (cfg == 1 && has_resid && N == 512 && K > 512 && wgs(256, 256) < 256) cfg = 22;      // (N = 512: only these layers were measured)
  else if (cfg == 4 && N > 64 && wgs(128, 64) < 200) cfg = 5;
  // less than one 64 x 64 workgroup per CU: every k-step is an exposed round trip -> two (tile 20) / four (tile 21, K >= 2048) slabs per
  // step.  r04, same box, one image per step: 2.38-2.40 -> 2.11-2.17 ms (tile 20) -> 2.08-2.12 ms; with more workgroups than CUs
  // (two images: 300) it loses 1-3 %
  if (cfg == 5 && wgs(64, 64) <= 256 && K >= 256) cfg = K >= 2048 ? 21 : 20;
  return cfg;
}
extern "C" int relnet_gemm_tile_count(void) { return GEMM_TILE_COUNT; }
extern "C" int relnet_gemm_pick_tile(int M, int N, int K, int batch, int out_dtype) { return pick_tile(M, N, K, batch, out_dtype, 0); }

template <int CONV>
static void launch_bf16(const GemmArgs& g, int batch, int out_dtype, hipStream_t s) {
  int cfg = g_force_tile;
  if (cfg <= 0 || cfg > GEMM_TILE_COUNT) {
    cfg = pick_tile(g.M, g.N, g.K, batch, out_dtype, g.resid != nullptr, CONV == 1 && g.Wf != nullptr);
    if ((cfg == 5 || cfg == 20 || cfg == 21) && CONV != 2 && splitk_ways(g.M, g.N, g.K, batch, g) > 1) cfg = 23;
  }
  switch (cfg) {
    case 1: launch_cfg<256, 256, 2, 4, CONV>(g, batch, out_dtype, s); break;
    case 2: launch_cfg<256, 128, 4, 2, CONV>(g, batch, out_dtype, s); break;
    case 3: launch_cfg<128, 128, 2, 2, CONV>(g, batch, out_dtype, s); break;
    case 4: launch_cfg<128, 64, 2, 2, CONV>(g, batch, out_dtype, s); break;
    case 5: launch_cfg<64, 64, 2, 2, CONV>(g, batch, out_dtype, s); break;
    case 20:
      if constexpr (CONV == 2) launch_cfg<64, 64, 2, 2, CONV>(g, batch, out_dtype, s);
      else launch_cfg<64, 64, 2, 2, CONV, 2>(g, batch, out_dtype, s);
      break;
    case 21:
      if constexpr (CONV == 2) launch_cfg<64, 64, 2, 2, CONV>(g, batch, out_dtype, s);
      else launch_cfg<64, 64, 2, 2, CONV, 4>(g, batch, out_dtype, s);
      break;
    // 23 (r06): split-K for the launches of at most one 64 x 64 workgroup per CU -- the one-image step (the reference's BATCH_IMAGES: 1
    // protocol, inference and training): res4 3x3 = 152 tiles x 36 k-slabs, fc_new_1 = 80 tiles x 196 k-slabs.  With one resident workgroup per
    // CU a k-step is one exposed L2 round trip whatever it carries (tiles 20 / 21 cut the steps by carrying 2 / 4 slabs); splitting the k-loop
    // over 2 - 8 workgroups per tile cuts the steps AND fills the idle CUs.  Partials in fp32 through the work area, summed in split order by
    // the last arriver, which also runs the bias / shortcut / ReLU epilogue: one launch, deterministic.
    case 23:
      if constexpr (CONV == 2) launch_cfg<64, 64, 2, 2, CONV>(g, batch, out_dtype, s);
      else {
        int ways = splitk_ways(g.M, g.N, g.K, batch, g);
        if (cfg == g_force_tile && ways == 1 && g_splitk != 1 && batch == 1 && !(g.N & 7) && !(g.ldc & 7) && g.K >= 128 && !((((uintptr_t)g.C) | ((uintptr_t)g.resid)) & 15)) ways = 2;   // forced tile (tests): always split
        launch_cfg<64, 64, 2, 2, CONV, 2>(g, batch, out_dtype, s, ways);
      }
      break;
    // 22 (r05): 192 x 128 (2 x 2 waves, 96 x 64 outputs each) for the 128-column 3x3 layers of res3: one column tile, so a taller tile only cuts
    // the filter re-reads (fill bytes per output -17 % against 128 x 128).  us per launch, tile 3 / 22: 54 images 215.8 / 200.7, 8 images 47.8 / 40.3
    // (forward), 52.0 / 43.5 (data gradient + mask).  Two k-slabs per barrier on this or the 128 x 128 / 160 x 128 tiles (one resident workgroup
    // per CU) were measured too: 56 - 84 us against 52 for res4 3x3 at 8 images -- slower, not kept
    case 22:
      if constexpr (CONV == 2) launch_cfg<128, 128, 2, 2, CONV>(g, batch, out_dtype, s);
      else launch_cfg<192, 128, 2, 2, CONV>(g, batch, out_dtype, s);
      break;
    case 6:
      if constexpr (CONV == 2) launch_cfg<256, 256, 2, 4, CONV>(g, batch, out_dtype, s);      // stem slabs are 64 deep
      else launch_ring<256, 256, 2, 4, CONV, 32, 4>(g, batch, out_dtype, s);
      break;
    case 7: launch_ring<256, 128, 4, 2, CONV, 64, 3>(g, batch, out_dtype, s); break;
    case 8: launch_ring<256, 256, 2, 4, CONV, 64, 2>(g, batch, out_dtype, s); break;
    case 9:
      if constexpr (CONV == 2) launch_cfg<256, 256, 2, 4, CONV>(g, batch, out_dtype, s);
      else launch_ring<256, 256, 2, 4, CONV, 32, 4, 1>(g, batch, out_dtype, s);
      break;
    case 10: launch_ring<256, 128, 4, 2, CONV, 64, 3, 1>(g, batch, out_dtype, s); break;
    case 11: launch_ring<256, 256, 2, 4, CONV, 64, 2, 0, 1>(g, batch, out_dtype, s); break;
    case 12: launch_ring<256, 256, 2, 4, CONV, 64, 2, 1>(g, batch, out_dtype, s); break;
    case 16: launch_ring<256, 256, 2, 4, CONV, 64, 2, 2>(g, batch, out_dtype, s); break;
    case 17:
      if constexpr (CONV == 1) {
        if (!g.resid && batch == 1 && g.cR == 3 && g.cS == 3 && g.cStride == 1 && g.cPad == g.cDil && g.cCin % 64 == 0 &&
            256 + 2 * (g.cW + 1) * g.cDil <= 384 && g.cImg % 8 == 0 && g.cHout == g.cH && g.cWout == g.cW &&
            (long)g.M / ((long)g.cH * g.cW) * g.cImg < (1L << 31) && (long)g.N * g.ldw < (1L << 31)) {
          launch_ring<256, 256, 2, 4, CONV, 64, 2, 4>(g, batch, out_dtype, s);
          break;
        }
      }
      launch_ring<256, 256, 2, 4, CONV, 64, 2>(g, batch, out_dtype, s);
      break;
    case 18:
    case 19:
      if constexpr (CONV != 2) {
        bool ok = !g.resid && g.N % 256 == 0 && g.K % 64 == 0 && (long)g.N * g.ldw * 2 < (1L << 32);
        if constexpr (CONV == 1) ok = ok && g.cCin % 64 == 0 && g.cR * g.cS <= 32 && (g.cR * g.cS == 1 || g_korder);
        if (ok && cfg != 18) { launch_ring<256, 256, 2, 4, CONV, 64, 2, 6>(g, batch, out_dtype, s); break; }
        if (ok) { launch_ring<256, 256, 2, 4, CONV, 64, 2, 5>(g, batch, out_dtype, s); break; }
      }
      launch_ring<256, 256, 2, 4, CONV, 64, 2>(g, batch, out_dtype, s);
      break;
    case 13:
      if (!launch_panel<CONV>(g, batch, out_dtype, s)) launch_cfg<256, 256, 2, 4, CONV>(g, batch, out_dtype, s);
      break;
    case 14:
      if (!launch_panel<CONV>(g, batch, out_dtype, s, true)) launch_cfg<256, 256, 2, 4, CONV>(g, batch, out_dtype, s);
      break;
    default:
      if (!launch_panel<CONV>(g, batch, out_dtype, s, true, true)) launch_cfg<256, 256, 2, 4, CONV>(g, batch, out_dtype, s);
      break;
  }
}

extern "C" int relnet_gemm_nt(const void* A, long lda, long strideA, const void* W, long ldw,
                              long strideW, void* C, long ldc, long strideC, const float* bias,
                              int bias_mode, const void* resid, int relu, int M, int N, int K,
                              int batch, int in_dtype, int out_dtype, void* stream) {
  RELNET_REQUIRE(A && W && C, "relnet_gemm_nt: null operand");
  RELNET_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "relnet_gemm_nt: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  RELNET_REQUIRE(bias_mode == 0 || bias, "relnet_gemm_nt: bias_mode=%d needs a bias vector", bias_mode);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.strideA = strideA; g.W = W; g.ldw = ldw; g.strideW = strideW; g.C = C; g.ldc = ldc;
  g.strideC = strideC; g.bias = bias; g.resid = resid; g.M = M; g.N = N; g.K = K; g.bias_mode = bias_mode; g.relu = relu;
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == RELNET_BF16) {
    RELNET_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0, "relnet_gemm_nt(bf16): K %% 64 and ld %% 8 required (K=%d lda=%ld ldw=%ld)", K, lda, ldw);
    launch_bf16<0>(g, batch, out_dtype, s);
  } else if (in_dtype == RELNET_F32) {
    RELNET_REQUIRE(K % 16 == 0 && lda % 4 == 0 && ldw % 4 == 0, "relnet_gemm_nt(f32): K %% 16 and ld %% 4 required (K=%d)", K);
    dim3 grid((N + 127) / 128, (M + 127) / 128, batch);
    if (out_dtype == RELNET_BF16) gemm_nt_f32_kernel<unsigned short><<<grid, 256, 0, s>>>(g);
    else gemm_nt_f32_kernel<float><<<grid, 256, 0, s>>>(g);
  } else {
    RELNET_REQUIRE(false, "relnet_gemm_nt: unknown in_dtype %d", in_dtype);
  }
  return check_launch("relnet_gemm_nt");
}

// C = (A W^T + resid) masked by (mask > 0): the data gradient through `x_next = relu(conv1x1(.) + x)` of a residual unit in ONE launch --
// `resid` = the gradient that arrives over the identity shortcut, `mask` = the saved forward activation whose ReLU the gradient passes
// (MXNet's autograd of Activation('relu') after broadcast_add, resnet_v1_101_rcnn_base.py: res*_relu).  bf16 in / out, batch 1;
// resid (may be NULL) and mask share C's layout.  Runs on the LDS-tiled kernel's MASK instantiations (the tile pick_tile would choose for a
// residual GEMM of this shape); replaces relnet_gemm_nt(resid) + relnet_relu_bwd (one more pass over three [M, N] maps).
template <int BM, int BN, int WM, int WN>
static void launch_cfg_mask(GemmArgs g, hipStream_t s) {
  const int ntile = (g.N + BN - 1) / BN;
  const bool swz = g_swizzle && ntile > 1 && (long)ntile * ((g.M + BM - 1) / BM) >= 16;
  g.n_loop = 1;
  dim3 grid(ntile, (g.M + BM - 1) / BM, 1);
  g.xcd_swizzle = swz ? 1 : 0;
  gemm_nt_bf16_kernel<BM, BN, WM, WN, unsigned short, 0, 1, true><<<grid, 64 * WM * WN, 0, s>>>(g);
}

extern "C" int relnet_gemm_nt_mask(const void* A, long lda, const void* W, long ldw, void* C, long ldc, const void* resid,
                                   const void* mask, int M, int N, int K, void* stream) {
  RELNET_REQUIRE(A && W && C && mask, "relnet_gemm_nt_mask: null operand");
  RELNET_REQUIRE(M > 0 && N > 0 && K > 0, "relnet_gemm_nt_mask: bad shape M=%d N=%d K=%d", M, N, K);
  RELNET_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0 && N % 8 == 0 && ldc % 8 == 0,
                 "relnet_gemm_nt_mask: K %% 64, N %% 8 and ld %% 8 required (N=%d K=%d lda=%ld ldw=%ld ldc=%ld)", N, K, lda, ldw, ldc);
  RELNET_REQUIRE((((uintptr_t)C | (uintptr_t)mask | (uintptr_t)resid) & 15) == 0, "relnet_gemm_nt_mask: C, resid and mask must be 16-byte aligned");
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.resid = resid; g.mask = mask;
  g.M = M; g.N = N; g.K = K; g.bias_mode = 0; g.relu = 3;
  hipStream_t s = (hipStream_t)stream;
  int cfg = g_force_tile;
  if (!((cfg >= 1 && cfg <= 5) || cfg == 22)) cfg = pick_tile(M, N, K, 1, RELNET_BF16, 1);
  switch (cfg) {
    case 1: launch_cfg_mask<256, 256, 2, 4>(g, s); break;
    case 22: launch_cfg_mask<192, 128, 2, 2>(g, s); break;
    case 4: launch_cfg_mask<128, 64, 2, 2>(g, s); break;
    case 5: case 20: case 21: launch_cfg_mask<64, 64, 2, 2>(g, s); break;
    default: launch_cfg_mask<128, 128, 2, 2>(g, s); break;
  }
  return check_launch("relnet_gemm_nt_mask");
}

// C (fp32) = A W^T with IEEE-half operands on the LDS-tiled kernel (tile: 2 = 256 x 128, 3 = 128 x 128, anything else = 256 x 128):
// the fp16 twin of relnet_gemm_nt(bf16 in, fp32 out), for the fp16-vs-bf16 rate measurement of tools/fp16_rate.py and its parity test.
extern "C" int relnet_gemm_nt_f16(const void* A, long lda, const void* W, long ldw, float* C, long ldc, int M, int N, int K, int tile,
                                  void* stream) {
  RELNET_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "relnet_gemm_nt_f16: bad operands");
  RELNET_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0, "relnet_gemm_nt_f16: K %% 64 and ld %% 8 required (K=%d lda=%ld ldw=%ld)", K, lda, ldw);
  GemmArgs g{};
  g.A = A; g.lda = lda; g.W = W; g.ldw = ldw; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.n_loop = 1;
  hipStream_t s = (hipStream_t)stream;
  if (tile == 3) {
    dim3 grid((N + 127) / 128, (M + 127) / 128, 1);
    gemm_nt_bf16_kernel<128, 128, 2, 2, float, 0, 1, false, true><<<grid, 256, 0, s>>>(g);
  } else {
    dim3 grid((N + 127) / 128, (M + 255) / 256, 1);
    gemm_nt_bf16_kernel<256, 128, 4, 2, float, 0, 1, false, true><<<grid, 512, 0, s>>>(g);
  }
  return check_launch("relnet_gemm_nt_f16");
}

// NHWC convolution as an implicit GEMM on the bf16 MFMA kernel (reference: the Convolution
// + BatchNorm(use_global_stats) + Activation triples of resnet_v1_101_rcnn_base.py:29-693, BN
// folded into weight/bias at load time; `resid` fuses the bottleneck's broadcast_add + ReLU).
//   in  [B, H, W, >=Cin] bf16 (pixel stride in_pix, image stride in_img, elements)
//   w   [Cout, R*S*Cin] bf16, k = (r*S + s)*Cin + ic
//   out [B*Hout*Wout, ldc] (bf16 or f32), resid same layout/dtype as out
extern "C" int relnet_conv2d_nhwc(const void* in, long in_pix, long in_img, const void* w,
                                  const float* bias, const void* resid, int relu, void* out, long ldc,
                                  int B, int H, int W, int Cin, int Cout, int R, int S, int stride,
                                  int dil, int pad, int out_dtype, void* stream) {
  RELNET_REQUIRE(in && w && out, "relnet_conv2d_nhwc: null operand");
  RELNET_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && dil > 0 && pad >= 0, "relnet_conv2d_nhwc: bad geometry");
  RELNET_REQUIRE(Cin % 64 == 0 && in_pix % 8 == 0 && in_img % 8 == 0, "relnet_conv2d_nhwc: Cin %% 64 == 0 and 16-byte aligned strides required (Cin=%d)", Cin);
  const int Hout = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  const int Wout = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  RELNET_REQUIRE(Hout > 0 && Wout > 0 && (long)B * Hout * Wout < (1L << 31), "relnet_conv2d_nhwc: empty or too large output");
  GemmArgs g{};
  g.A = in; g.lda = in_pix; g.strideA = 0; g.W = w; g.ldw = (long)R * S * Cin; g.strideW = 0;
  g.C = out; g.ldc = ldc; g.strideC = 0; g.bias = bias; g.resid = resid;
  g.M = B * Hout * Wout; g.N = Cout; g.K = R * S * Cin; g.bias_mode = bias ? 1 : 0; g.relu = relu;
  g.cH = H; g.cW = W; g.cCin = Cin; g.cHout = Hout; g.cWout = Wout; g.cR = R; g.cS = S;
  g.cStride = stride; g.cDil = dil; g.cPad = pad; g.cPix = in_pix; g.cImg = in_img;
  launch_bf16<1>(g, 1, out_dtype, (hipStream_t)stream);
  return check_launch("relnet_conv2d_nhwc");
}

// The same convolution with float32 operands on the exact-fp32 MFMA kernel (v_mfma_f32_32x32x2f32: bit-wise an fmaf chain per output
// element) -- the float32 PARITY path of every convolution of the graph (backbone, RPN head, conv_new_1, FPN neck), in place of a
// library call.  in [B,H,W,>=Cin] fp32 (pixel / image strides in elements, multiples of 4), w [Cout][R*S*Cin] fp32, out / resid
// [B*Hout*Wout][ldc] fp32.  Cin % 16 == 0 (the 3-channel stem runs on an image zero-padded to 16 channels).  relu: 0 none, 1 ReLU,
// 2 = resid is a ReLU mask (data gradients of the float32 training path).
extern "C" int relnet_conv2d_nhwc_f32(const float* in, long in_pix, long in_img, const float* w, const float* bias, const float* resid,
                                      int relu, float* out, long ldc, int B, int H, int W, int Cin, int Cout, int R, int S, int stride,
                                      int dil, int pad, void* stream) {
  RELNET_REQUIRE(in && w && out, "relnet_conv2d_nhwc_f32: null operand");
  RELNET_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && dil > 0 && pad >= 0, "relnet_conv2d_nhwc_f32: bad geometry");
  RELNET_REQUIRE(Cin % 16 == 0 && in_pix % 4 == 0 && in_img % 4 == 0 && (((uintptr_t)in | (uintptr_t)w) & 15) == 0,
                 "relnet_conv2d_nhwc_f32: Cin %% 16 == 0 and 16-byte aligned operands / strides required (Cin=%d in_pix=%ld)", Cin, in_pix);
  const int Hout = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  const int Wout = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  RELNET_REQUIRE(Hout > 0 && Wout > 0 && (long)B * Hout * Wout < (1L << 31), "relnet_conv2d_nhwc_f32: empty or too large output");
  GemmArgs g{};
  g.A = in; g.lda = in_pix; g.strideA = 0; g.W = w; g.ldw = (long)R * S * Cin; g.strideW = 0;
  g.C = out; g.ldc = ldc; g.strideC = 0; g.bias = bias; g.resid = resid;
  g.M = B * Hout * Wout; g.N = Cout; g.K = R * S * Cin; g.bias_mode = bias ? 1 : 0; g.relu = relu;
  g.cH = H; g.cW = W; g.cCin = Cin; g.cHout = Hout; g.cWout = Wout; g.cR = R; g.cS = S;
  g.cStride = stride; g.cDil = dil; g.cPad = pad; g.cPix = in_pix; g.cImg = in_img;
  dim3 grid((Cout + 127) / 128, (g.M + 127) / 128, 1);
  gemm_nt_f32_kernel<float, true><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_conv2d_nhwc_f32");
}

// W [N][K] bf16 -> MFMA fragment order for gemm_panelw_kernel (N % 32 == 0, K % 16 == 0); done once at model load
extern "C" int relnet_pack_w_frag(const void* w, long ldw, void* out, int N, int K, void* stream) {
  RELNET_REQUIRE(w && out && N > 0 && K > 0 && N % 32 == 0 && K % 16 == 0 && ldw % 8 == 0, "relnet_pack_w_frag: bad arguments (N=%d K=%d)", N, K);
  const long total = (long)(N / 32) * (K / 16) * 64;
  pack_w_frag_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)w, ldw, (uint4*)out, N, K);
  return check_launch("relnet_pack_w_frag");
}

// relnet_conv2d_nhwc with the optional fragment-order copy of the weights (w_frag may be NULL)
extern "C" int relnet_conv2d_nhwc_wf(const void* in, long in_pix, long in_img, const void* w, const void* w_frag,
                                     const float* bias, const void* resid, int relu, void* out, long ldc,
                                     int B, int H, int W, int Cin, int Cout, int R, int S, int stride,
                                     int dil, int pad, int out_dtype, void* stream) {
  RELNET_REQUIRE(in && w && out, "relnet_conv2d_nhwc: null operand");
  RELNET_REQUIRE(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && dil > 0 && pad >= 0, "relnet_conv2d_nhwc: bad geometry");
  RELNET_REQUIRE(Cin % 64 == 0 && in_pix % 8 == 0 && in_img % 8 == 0, "relnet_conv2d_nhwc: Cin %% 64 == 0 and 16-byte aligned strides required (Cin=%d)", Cin);
  const int Hout = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1;
  const int Wout = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
  RELNET_REQUIRE(Hout > 0 && Wout > 0 && (long)B * Hout * Wout < (1L << 31), "relnet_conv2d_nhwc: empty or too large output");
  GemmArgs g{};
  g.A = in; g.lda = in_pix; g.strideA = 0; g.W = w; g.ldw = (long)R * S * Cin; g.strideW = 0;
  g.C = out; g.ldc = ldc; g.strideC = 0; g.bias = bias; g.resid = resid;
  g.M = B * Hout * Wout; g.N = Cout; g.K = R * S * Cin; g.bias_mode = bias ? 1 : 0; g.relu = relu;
  g.cH = H; g.cW = W; g.cCin = Cin; g.cHout = Hout; g.cWout = Wout; g.cR = R; g.cS = S;
  g.cStride = stride; g.cDil = dil; g.cPad = pad; g.cPix = in_pix; g.cImg = in_img;
  g.Wf = w_frag;
  launch_bf16<1>(g, 1, out_dtype, (hipStream_t)stream);
  return check_launch("relnet_conv2d_nhwc");
}

// ---------------------------------------------------------------------------------------
// Stem: conv1 7x7 / stride 2 / pad 3, Cin = 3 (resnet_v1_101_rcnn_base.py:30-31) as an implicit GEMM
// on the same MFMA kernel.  Cin = 3 cannot feed 16-byte fragment loads, so the image is first
// repacked (relnet_stem_pack_input) to zero-padded NHWC4 bf16 [B][H+6 (+1)][W+8][4]: every 7-tap row
// of a window is then 28 contiguous elements (padded to 32 with the next pixel, whose weights are
// zero) at an 8-byte aligned address and no bounds checks are needed.  K = 8 tap rows x 32 = 256
// (row 7 and channel 3 carry zero weights): w [Cout][256], k = ty*32 + tx*4 + c.
// ---------------------------------------------------------------------------------------
namespace relnet {
struct PackArgs {
  const void* in;            // [B, 3, H, W] fp32 or bf16 (NCHW, contiguous)
  unsigned short* out;       // [B, Hp, Wp, 4] bf16
  int B, H, W, Hp, Wp, pad, in_bf16;
};
__global__ __launch_bounds__(256) void stem_pack_kernel(PackArgs g) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)g.B * g.Hp * g.Wp;
  if (t >= total) return;
  const int xp = (int)(t % g.Wp);
  const int yp = (int)((t / g.Wp) % g.Hp);
  const int b = (int)(t / ((long)g.Wp * g.Hp));
  const int x = xp - g.pad, y = yp - g.pad;
  float v[3] = {0.f, 0.f, 0.f};
  if (x >= 0 && x < g.W && y >= 0 && y < g.H) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const long o = (((long)b * 3 + c) * g.H + y) * g.W + x;
      v[c] = g.in_bf16 ? bf2f(((const unsigned short*)g.in)[o]) : ((const float*)g.in)[o];
    }
  }
  *(uint2*)(g.out + t * 4) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], 0.f));
}
}  // namespace relnet

extern "C" int relnet_stem_pack_input(const void* in, void* out, int B, int H, int W, int Hp, int Wp, int pad,
                                      int in_dtype, void* stream) {
  RELNET_REQUIRE(in && out && B > 0 && Hp >= H + 2 * pad && Wp >= W + 2 * pad, "relnet_stem_pack_input: bad arguments");
  PackArgs g{in, (unsigned short*)out, B, H, W, Hp, Wp, pad, in_dtype == RELNET_BF16};
  const long total = (long)B * Hp * Wp;
  stem_pack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_stem_pack_input");
}

// packed [B][Hp][Wp][4] -> out [B*Hout*Wout, ldc] = relu?(conv7x7/2 + bias); Hout = (H+6-7)/2+1 etc.
extern "C" int relnet_stem_conv7(const void* packed, const void* w256, const float* bias, int relu, void* out,
                                 long ldc, int B, int Hp, int Wp, int Hout, int Wout, int Cout, int out_dtype,
                                 void* stream) {
  RELNET_REQUIRE(packed && w256 && out, "relnet_stem_conv7: null operand");
  RELNET_REQUIRE(2 * (Hout - 1) + 8 <= Hp && 2 * (Wout - 1) + 8 <= Wp && Wp % 2 == 0, "relnet_stem_conv7: padded image too small (Hp=%d Wp=%d)", Hp, Wp);
  GemmArgs g{};
  g.A = packed; g.W = w256; g.ldw = 256; g.C = out; g.ldc = ldc; g.bias = bias; g.bias_mode = bias ? 1 : 0; g.relu = relu;
  g.M = B * Hout * Wout; g.N = Cout; g.K = 256;
  g.cHout = Hout; g.cWout = Wout; g.cW = Wp; g.cImg = (long)Hp * Wp * 4;
  launch_bf16<2>(g, 1, out_dtype, (hipStream_t)stream);
  return check_launch("relnet_stem_conv7");
}
