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
};

template <typename TOUT> __device__ __forceinline__ void store_out(TOUT* p, float v);
template <> __device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <> __device__ __forceinline__ void store_out<unsigned short>(unsigned short* p, float v) { *p = f2bf(v); }
template <typename TOUT> __device__ __forceinline__ float load_out(const TOUT* p);
template <> __device__ __forceinline__ float load_out<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_out<unsigned short>(const unsigned short* p) { return bf2f(*p); }

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
      if (R) v += load_out<TOUT>(R + (long)row * g.ldc + col);
      if (g.relu) v = fmaxf(v, 0.f);
      store_out<TOUT>(C + (long)row * g.ldc + col, v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// bf16 in, fp32 accumulate.  BM x BN workgroup tile, TM x TN MFMA tiles per wave.
// ---------------------------------------------------------------------------------------
template <int BM, int BN, typename TOUT>
__global__ __launch_bounds__(256) void gemm_nt_bf16_kernel(GemmArgs g) {
  constexpr int BK = 64;
  constexpr int TM = BM / 64, TN = BN / 64;           // 32x32 tiles per wave per dim
  constexpr int A_CHUNKS = BM * 8 / 256, B_CHUNKS = BN * 8 / 256;   // 16-B chunks per thread
  constexpr int STAGE = (BM + BN) * BK * 2;            // bytes per pipeline stage
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];
  auto ldsA = [&](int buf) { return lds + buf * STAGE; };
  auto ldsB = [&](int buf) { return lds + buf * STAGE + BM * BK * 2; };

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  const unsigned short* A = (const unsigned short*)g.A + (long)blockIdx.z * g.strideA;
  const unsigned short* W = (const unsigned short*)g.W + (long)blockIdx.z * g.strideW;
  TOUT* C = (TOUT*)g.C + (long)blockIdx.z * g.strideC;
  const TOUT* R = g.resid ? (const TOUT*)g.resid + (long)blockIdx.z * g.strideC : nullptr;

  uint4 ra[A_CHUNKS], rb[B_CHUNKS];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      const int gr = m0 + row;
      ra[i] = (gr < g.M) ? *(const uint4*)(A + (long)gr * g.lda + k0 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      const int gr = n0 + row;
      rb[i] = (gr < g.N) ? *(const uint4*)(W + (long)gr * g.ldw + k0 + ch * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      *(uint4*)(ldsA(buf) + row * 128 + ((ch ^ (row & 7)) << 4)) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_CHUNKS; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      *(uint4*)(ldsB(buf) + row * 128 + ((ch ^ (row & 7)) << 4)) = rb[i];
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = g.K / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[TM], bfr[TN];
      const int ch = 2 * kk + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int row = wr * (BM / 2) + i * 32 + (lane & 31);
        af[i] = *(const bf16x8*)(ldsA(buf) + row * 128 + ((ch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int row = wc * (BN / 2) + j * 32 + (lane & 31);
        bfr[j] = *(const bf16x8*)(ldsB(buf) + row * 128 + ((ch ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      epilogue_tile<TOUT>(g, C, R, acc[i][j], m0 + wr * (BM / 2) + i * 32, n0 + wc * (BN / 2) + j * 32, lane);
}

// ---------------------------------------------------------------------------------------
// f32 in / f32 accumulate (exact fp32 MFMA, bit-wise an fmaf chain).  128x128 tile.
// ---------------------------------------------------------------------------------------
template <typename TOUT>
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
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int ra = m0 + i * 32 + (lane & 31); ra = ra < g.M ? ra : g.M - 1;
    int rb = n0 + i * 32 + (lane & 31); rb = rb < g.N ? rb : g.N - 1;
    pa[i] = A + (long)ra * g.lda + half * 8;
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
  auto load = [&](int kb, float4 (&x)[2][2], float4 (&y)[2][2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      x[i][0] = *(const float4*)(pa[i] + kb * 16);
      x[i][1] = *(const float4*)(pa[i] + kb * 16 + 4);
      y[i][0] = *(const float4*)(pb[i] + kb * 16);
      y[i][1] = *(const float4*)(pb[i] + kb * 16 + 4);
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

extern "C" int relnet_gemm_nt(const void* A, long lda, long strideA, const void* W, long ldw,
                              long strideW, void* C, long ldc, long strideC, const float* bias,
                              int bias_mode, const void* resid, int relu, int M, int N, int K,
                              int batch, int in_dtype, int out_dtype, void* stream) {
  RELNET_REQUIRE(A && W && C, "relnet_gemm_nt: null operand");
  RELNET_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0, "relnet_gemm_nt: bad shape M=%d N=%d K=%d batch=%d", M, N, K, batch);
  RELNET_REQUIRE(bias_mode == 0 || bias, "relnet_gemm_nt: bias_mode=%d needs a bias vector", bias_mode);
  GemmArgs g{A, lda, strideA, W, ldw, strideW, C, ldc, strideC, bias, resid, M, N, K, bias_mode, relu};
  hipStream_t s = (hipStream_t)stream;
  if (in_dtype == RELNET_BF16) {
    RELNET_REQUIRE(K % 64 == 0 && lda % 8 == 0 && ldw % 8 == 0, "relnet_gemm_nt(bf16): K %% 64 and ld %% 8 required (K=%d lda=%ld ldw=%ld)", K, lda, ldw);
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    if (tiles128 >= 200) {
      dim3 grid((N + 127) / 128, (M + 127) / 128, batch);
      if (out_dtype == RELNET_BF16) gemm_nt_bf16_kernel<128, 128, unsigned short><<<grid, 256, 0, s>>>(g);
      else gemm_nt_bf16_kernel<128, 128, float><<<grid, 256, 0, s>>>(g);
    } else {
      dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
      if (out_dtype == RELNET_BF16) gemm_nt_bf16_kernel<64, 64, unsigned short><<<grid, 256, 0, s>>>(g);
      else gemm_nt_bf16_kernel<64, 64, float><<<grid, 256, 0, s>>>(g);
    }
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
