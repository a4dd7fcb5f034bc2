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
  // implicit-GEMM convolution (CONV kernels only): A is an NHWC image batch, row m of the
  // GEMM is output pixel (b, oy, ox), k = (r*S + s)*Cin + ic; W is [Cout][R][S][Cin].
  int cH, cW, cCin, cHout, cWout, cR, cS, cStride, cDil, cPad;
  long cPix, cImg;         // element strides between pixels / images of the input
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
template <int BM, int BN, typename TOUT, bool CONV>
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
  // implicit im2col: per-thread output-pixel coordinates of its A rows (fixed over k)
  const unsigned short* cbase[A_CHUNKS];
  int ciy[A_CHUNKS], cix[A_CHUNKS];
  if constexpr (CONV) {
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      const int gr = m0 + ((tid + 256 * i) >> 3);
      const int hw = g.cHout * g.cWout;
      const int b = gr / hw, rem = gr - b * hw;
      const int oy = rem / g.cWout, ox = rem - oy * g.cWout;
      cbase[i] = A + (long)b * g.cImg;
      ciy[i] = (gr < g.M) ? oy * g.cStride - g.cPad : -(1 << 28);      // row >= M: never in bounds
      cix[i] = ox * g.cStride - g.cPad;
    }
  }
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    int tr = 0, ts = 0, ic0 = k0;
    if constexpr (CONV) {
      const int tap = k0 / g.cCin;
      ic0 = k0 - tap * g.cCin;
      tr = tap / g.cS; ts = tap - tr * g.cS;
    }
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
      const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
      if constexpr (CONV) {
        const int iy = ciy[i] + tr * g.cDil, ix = cix[i] + ts * g.cDil;
        const bool ok = (iy >= 0) && (iy < g.cH) && (ix >= 0) && (ix < g.cW);
        ra[i] = ok ? *(const uint4*)(cbase[i] + ((long)iy * g.cW + ix) * g.cPix + ic0 + ch * 8) : make_uint4(0, 0, 0, 0);
      } else {
        const int gr = m0 + row;
        ra[i] = (gr < g.M) ? *(const uint4*)(A + (long)gr * g.lda + k0 + ch * 8) : make_uint4(0, 0, 0, 0);
      }
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

template <bool CONV>
static void launch_bf16(const GemmArgs& g, int batch, int out_dtype, hipStream_t s) {
  const int M = g.M, N = g.N;
  const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (N <= 64) {
    dim3 grid((N + 63) / 64, (M + 127) / 128, batch);
    if (out_dtype == RELNET_BF16) gemm_nt_bf16_kernel<128, 64, unsigned short, CONV><<<grid, 256, 0, s>>>(g);
    else gemm_nt_bf16_kernel<128, 64, float, CONV><<<grid, 256, 0, s>>>(g);
  } else if (tiles128 >= 200) {
    dim3 grid((N + 127) / 128, (M + 127) / 128, batch);
    if (out_dtype == RELNET_BF16) gemm_nt_bf16_kernel<128, 128, unsigned short, CONV><<<grid, 256, 0, s>>>(g);
    else gemm_nt_bf16_kernel<128, 128, float, CONV><<<grid, 256, 0, s>>>(g);
  } else {
    dim3 grid((N + 63) / 64, (M + 63) / 64, batch);
    if (out_dtype == RELNET_BF16) gemm_nt_bf16_kernel<64, 64, unsigned short, CONV><<<grid, 256, 0, s>>>(g);
    else gemm_nt_bf16_kernel<64, 64, float, CONV><<<grid, 256, 0, s>>>(g);
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
    launch_bf16<false>(g, batch, out_dtype, s);
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
  launch_bf16<true>(g, 1, out_dtype, (hipStream_t)stream);
  return check_launch("relnet_conv2d_nhwc");
}
