// Backward of the object-relation module (training: SURVEY.md section 8 rows A7 / A10 / A13 -- MXNet derives
// it by autograd from symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py:85-151;
// there is no backward source in the reference to follow, the math is the adjoint of the forward kernels).
//
// Forward (per image b, head h):  L = log(max(G,1e-6)) + s Q K^T,  S = softmax_keys(L),  Y = S VW + bout,
//                                 VW = F_K Wout_h^T,  G = relu(E Wp^T + bp)
// Backward, given dY:
//   D_i   = sum_d dY[i,d] (Y[i,d] - bout[d])                 (= sum_m S dS)
//   dS    = dY VW^T,   dL = S * (dS - D)
//   dQ    = s dL K,    dK = s dL^T Q,    dVW = S^T dY
//   dpre  = dL / G  where G > 1e-6 (else 0);  dWp = dpre^T E,  dbp = sum dpre       (geometry kernel)
//
//  relation_attention_bwd_q_kernel   one wavefront = 32 queries of one (image, head): recomputes the logits
//        (two passes over the keys: row statistics, then probabilities), writes S and dL ([B,H,N,Mpad] fp32)
//        and dQ.  Same swapped-MFMA layout as the forward kernel: a lane owns one query column.
//  relation_attention_bwd_kv_kernel  one wavefront = 32 keys of one (image, head): dVW = S^T dY and
//        dK = s dL^T Q as two more MFMA products over the stored S / dL; the A operands are the transposed
//        copies dY^T / Q^T ([B][H*64][Npad], queries contiguous).
//  geometry_bias_bwd_kernel          one wavefront = one query row: E is recomputed on the fly (one sin or cos
//        per lane), dpre comes from dL and the forward's log G, dWp accumulates in a 32x32x2 fp32 MFMA tile.
//  transpose_2d_kernel               [rows][cols] -> [cols][rows] through an LDS tile (operand layouts above).
#include "common.h"

namespace relnet {

enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

// ---------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void transpose_2d_kernel(const T* in, long in_ld, long in_bs, T* out, long out_ld,
                                                           long out_bs, int rows, int cols) {
  __shared__ T tile[64][65];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const T* src = in + (long)b * in_bs;
  T* dst = out + (long)b * out_bs;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 64 x 4
  for (int r = ty; r < 64; r += 4)
    if (r0 + r < rows && c0 + tx < cols) tile[r][tx] = src[(long)(r0 + r) * in_ld + c0 + tx];
  __syncthreads();
  for (int c = ty; c < 64; c += 4)
    if (c0 + c < cols && r0 + tx < rows) dst[(long)(c0 + c) * out_ld + r0 + tx] = tile[tx][c];
}

// bf16 fast path: 16-byte global loads and stores (8 elements per lane); requires 16-byte aligned rows on both
// sides.  Tile 64 x 64; the LDS image is [row][col] with a 66-element pitch (33 words: conflict-free column reads).
__global__ __launch_bounds__(256) void transpose_2d_bf16_vec_kernel(const unsigned short* in, long in_ld, long in_bs,
                                                                    unsigned short* out, long out_ld, long out_bs,
                                                                    int rows, int cols) {
  __shared__ unsigned short tile[64][66];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const unsigned short* src = in + (long)b * in_bs;
  unsigned short* dst = out + (long)b * out_bs;
  const int t = threadIdx.x;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int r = pass * 32 + (t >> 3), cc = (t & 7) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (r0 + r < rows && c0 + cc < cols) v = *(const uint4*)(src + (long)(r0 + r) * in_ld + c0 + cc);   // cols % 8 == 0
    unsigned int* p = (unsigned int*)&tile[r][cc];
    p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w;
  }
  __syncthreads();
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int c = pass * 32 + (t >> 3), rr = (t & 7) * 8;          // output row = input column c, 8 input rows rr..rr+7
    if (c0 + c < cols && r0 + rr < rows) {
      unsigned int w[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) w[k] = (unsigned int)tile[rr + 2 * k][c] | ((unsigned int)tile[rr + 2 * k + 1][c] << 16);
      *(uint4*)(dst + (long)(c0 + c) * out_ld + r0 + rr) = make_uint4(w[0], w[1], w[2], w[3]);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
struct AttnBwdArgs {
  const void* q; long q_ld, q_bs;          // [B][N][.. h*64+d ..]
  const void* k; long k_ld, k_bs;          // [B][M][.. h*64+d ..]
  const void* kt; long kt_ld, kt_bs;       // K^T  [B][H*64][Mpad] (keys contiguous, zero padded)
  const void* vw; long vw_ld, vw_bs;       // VW   [B][M][H*64]    (NOT transposed)
  const float* bias; long bias_bs;         // [B][H][N][Mpad] fp32 log G
  const void* dy; long dy_ld, dy_bs;       // [B][N][H*64] gradient of the module output
  const void* y; long y_ld, y_bs;          // [B][N][H*64] forward output (incl. bout)
  const float* bout;                       // [H*64] or nullptr
  const void* qt; long qt_ld, qt_bs;       // Q^T  [B][H*64][Npad] (queries contiguous, zero padded)
  const void* dyt; long dyt_ld, dyt_bs;    // dY^T [B][H*64][Npad]
  float* prob; float* dlog;                // [B][H][N][Mpad] fp32 (written by the q kernel, read by the others)
  float* dq; float* dk; float* dvw;        // fp32 [B][N][H*64], [B][M][H*64], [B][M][H*64]
  int B, H, N, M, Mpad;
  float scale;
  const int* key_count;                    // optional [B]: keys >= key_count[b] of image b are masked (as in the forward kernels)
};

// exp of the softmax recompute: libm expf on the float32 parity path; the hardware exponential (v_exp_f32, ~1 ulp) on the bf16 training path,
// where a wavefront evaluates ~34 of them per key tile and the ~40-instruction libm sequence made the backward kernels VALU bound
// (round 5: the small-N kernel with one wavefront per SIMD spent ~10 of its 30 us per workgroup in expf).
template <typename T> __device__ __forceinline__ float sm_exp(float x) {
  if constexpr (sizeof(T) == 2) return __expf(x);
  else return expf(x);
}

template <typename T> struct Frag;
template <> struct Frag<unsigned short> {           // 64 contraction values of one row as 4 bf16x8 fragments
  bf16x8 f[4];
  __device__ __forceinline__ void load(const unsigned short* row, int half) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = *(const bf16x8*)(row + 16 * kk + 8 * half);
  }
  __device__ __forceinline__ float get(int kk, int j) const { return bf2f((unsigned short)f[kk][j]); }
};
template <> struct Frag<float> {                    // fp32: this half-wave's 32 of the 64 values
  float f[32];
  __device__ __forceinline__ void load(const float* row, int half) {
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      const float4 v = *(const float4*)(row + half * 32 + s);
      f[s] = v.x; f[s + 1] = v.y; f[s + 2] = v.z; f[s + 3] = v.w;
    }
  }
};

// C^T[row][col=this lane's column] over 64 contraction values: A = `arow` (row l31 of the A matrix), B = frag
template <typename T>
__device__ __forceinline__ f32x16 dot64(const T* arow, const Frag<T>& bf, int half) {
  f32x16 s;
#pragma unroll
  for (int r = 0; r < 16; ++r) s[r] = 0.f;
  if constexpr (sizeof(T) == 2) {
    bf16x8 af[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) af[kk] = *(const bf16x8*)(arow + 16 * kk + 8 * half);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk], bf.f[kk], s, 0, 0, 0);
  } else {
    const float* ar = (const float*)arow + half * 32;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 v = *(const float4*)(ar + 4 * c4);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, bf.f[4 * c4 + 0], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, bf.f[4 * c4 + 1], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, bf.f[4 * c4 + 2], s, 0, 0, 0);
      s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, bf.f[4 * c4 + 3], s, 0, 0, 0);
    }
  }
  return s;
}

// o[d] += A[32 d + l31][x0 ...] * p   where p holds this lane's 16 entries (slot r <-> x0 + 8 (r>>2) + 4 half + (r&3)),
// `abase` = A + x0 + 4 * half for row 0 of the 64-row block, `ald` its row stride (the forward's PV product).
template <typename T>
__device__ __forceinline__ void acc_pv(f32x16 (&o)[2], const T* abase, long ald, int l31, const f32x16& p) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pf;
      unsigned int* pw = (unsigned int*)&pf;
#pragma unroll
      for (int t = 0; t < 4; ++t) pw[t] = pack_bf16x2(p[8 * ks + 2 * t], p[8 * ks + 2 * t + 1]);
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const unsigned short* ar = (const unsigned short*)abase + (long)(32 * d + l31) * ald + 16 * ks;
        bf16x8 af;
        *(uint2*)&af = *(const uint2*)(ar);
        *((uint2*)&af + 1) = *(const uint2*)(ar + 8);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf, o[d], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const float* ar = (const float*)abase + (long)(32 * d + l31) * ald;
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 v = *(const float4*)(ar + 8 * gq);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, p[4 * gq + 0], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, p[4 * gq + 1], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, p[4 * gq + 2], o[d], 0, 0, 0);
        o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, p[4 * gq + 3], o[d], 0, 0, 0);
      }
    }
  }
}

// store o (rows = 64 channels of head h, col = this lane's row index) * mul into out[row_index][h*64 + ...] fp32
__device__ __forceinline__ void store_rows(float* orow, const f32x16 (&o)[2], int half, float mul) {
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int dv = 32 * d + 8 * gq + 4 * half;
      *(float4*)(orow + dv) = make_float4(o[d][4 * gq] * mul, o[d][4 * gq + 1] * mul, o[d][4 * gq + 2] * mul, o[d][4 * gq + 3] * mul);
    }
}

// ... the same rows rounded to bf16 (the [dQ | dK | dVW] operand of the projection backward)
__device__ __forceinline__ void store_rows_bf16(unsigned short* orow, const f32x16 (&o)[2], int half, float mul) {
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int dv = 32 * d + 8 * gq + 4 * half;
      *(uint2*)(orow + dv) = make_uint2(pack_bf16x2(o[d][4 * gq] * mul, o[d][4 * gq + 1] * mul), pack_bf16x2(o[d][4 * gq + 2] * mul, o[d][4 * gq + 3] * mul));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void relation_attention_bwd_q_kernel(AttnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int qt = blockIdx.x * 4 + wave;
  const int h = blockIdx.y, b = blockIdx.z;
  if (qt * 32 >= a.N) return;
  const int q = qt * 32 + l31;
  const int qc = q < a.N ? q : a.N - 1;

  const T* Qr = (const T*)a.q + (long)b * a.q_bs + (long)qc * a.q_ld + h * 64;
  const T* dYr = (const T*)a.dy + (long)b * a.dy_bs + (long)qc * a.dy_ld + h * 64;
  const T* Yr = (const T*)a.y + (long)b * a.y_bs + (long)qc * a.y_ld + h * 64;
  const T* Kb = (const T*)a.k + (long)b * a.k_bs + h * 64;
  const T* VWb = (const T*)a.vw + (long)b * a.vw_bs + h * 64;
  const T* KTb = (const T*)a.kt + (long)b * a.kt_bs + (long)(h * 64) * a.kt_ld;
  const float* Bq = a.bias + (long)b * a.bias_bs + ((long)h * a.N + qc) * a.Mpad;
  float* Pq = a.prob + (((long)b * a.H + h) * a.N + qc) * a.Mpad;
  float* Lq = a.dlog + (((long)b * a.H + h) * a.N + qc) * a.Mpad;

  Frag<T> qf, dyf;
  qf.load(Qr, half);
  dyf.load(dYr, half);
  // D = sum_d dY (Y - bout) over this lane's half of the 64 channels, then across the two halves
  float D = 0.f;
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const bf16x8 yv = *(const bf16x8*)((const unsigned short*)Yr + 16 * kk + 8 * half);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int d = 16 * kk + 8 * half + j;
        D += dyf.get(kk, j) * (bf2f((unsigned short)yv[j]) - (a.bout ? a.bout[h * 64 + d] : 0.f));
      }
    }
  } else {
#pragma unroll
    for (int s = 0; s < 32; ++s) {
      const int d = half * 32 + s;
      D += dyf.f[s] * (((const float*)Yr)[d] - (a.bout ? a.bout[h * 64 + d] : 0.f));
    }
  }
  D += __shfl_xor(D, 32);

  const int nkt = (a.M + 31) / 32;
  const int Mb = a.key_count ? min(max(a.key_count[b], 1), a.M) : a.M;
  // pass 1: row maximum and normaliser
  float m_run = -INFINITY, l_run = 0.f;
  for (int kt = 0; kt < nkt; ++kt) {
    const int key0 = kt * 32;
    int kr = key0 + l31; kr = kr < a.M ? kr : a.M - 1;
    f32x16 s = dot64<T>(Kb + (long)kr * a.k_ld, qf, half);
    float tmax = -INFINITY;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kbase = key0 + 8 * gq + 4 * half;
      const float4 bv = *(const float4*)(Bq + kbase);
      const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = bb[e] + a.scale * s[4 * gq + e];
        v = (kbase + e < Mb) ? v : -INFINITY;
        s[4 * gq + e] = v;
        tmax = fmaxf(tmax, v);
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) psum += sm_exp<T>(s[r] - m_new);
    l_run = l_run * sm_exp<T>(m_run - m_new) + psum;
    m_run = m_new;
  }
  const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));

  // pass 2: S, dS, dL, dQ
  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  for (int kt = 0; kt < nkt; ++kt) {
    const int key0 = kt * 32;
    int kr = key0 + l31; kr = kr < a.M ? kr : a.M - 1;
    f32x16 s = dot64<T>(Kb + (long)kr * a.k_ld, qf, half);
    const f32x16 ds = dot64<T>(VWb + (long)kr * a.vw_ld, dyf, half);
    f32x16 dl;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kbase = key0 + 8 * gq + 4 * half;
      const float4 bv = *(const float4*)(Bq + kbase);
      const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
      float pv[4], lv[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        const float v = bb[e] + a.scale * s[r];
        pv[e] = (kbase + e < Mb) ? sm_exp<T>(v - m_run) * inv : 0.f;
        lv[e] = pv[e] * (ds[r] - D);
        dl[r] = lv[e];
      }
      if (q < a.N) {
        *(float4*)(Pq + kbase) = make_float4(pv[0], pv[1], pv[2], pv[3]);
        *(float4*)(Lq + kbase) = make_float4(lv[0], lv[1], lv[2], lv[3]);
      }
    }
    acc_pv<T>(o, KTb + key0 + 4 * half, a.kt_ld, l31, dl);
  }
  if (q < a.N) store_rows(a.dq + ((long)b * a.N + q) * (a.H * 64) + h * 64, o, half, a.scale);
}

template <typename T>
__global__ __launch_bounds__(256) void relation_attention_bwd_kv_kernel(AttnBwdArgs a, int Npad) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int kt = blockIdx.x * 4 + wave;
  const int h = blockIdx.y, b = blockIdx.z;
  if (kt * 32 >= a.M) return;
  const int key = kt * 32 + l31;                       // < Mpad: the S / dL rows are padded
  const float* Pb = a.prob + (((long)b * a.H + h) * a.N) * a.Mpad + key;
  const float* Lb = a.dlog + (((long)b * a.H + h) * a.N) * a.Mpad + key;
  const T* DYT = (const T*)a.dyt + (long)b * a.dyt_bs + (long)(h * 64) * a.dyt_ld;
  const T* QT = (const T*)a.qt + (long)b * a.qt_bs + (long)(h * 64) * a.qt_ld;
  f32x16 ov[2], ok[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) { ov[d][r] = 0.f; ok[d][r] = 0.f; }
  const int nqt = (a.N + 31) / 32;
  for (int t = 0; t < nqt; ++t) {
    const int q0 = t * 32;
    f32x16 p, dl;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = q0 + 8 * (r >> 2) + 4 * half + (r & 3);
      const bool okq = qq < a.N;
      p[r] = okq ? Pb[(long)qq * a.Mpad] : 0.f;
      dl[r] = okq ? Lb[(long)qq * a.Mpad] : 0.f;
    }
    acc_pv<T>(ov, DYT + q0 + 4 * half, a.dyt_ld, l31, p);
    acc_pv<T>(ok, QT + q0 + 4 * half, a.qt_ld, l31, dl);
  }
  if (key < a.M) {
    store_rows(a.dvw + ((long)b * a.M + key) * (a.H * 64) + h * 64, ov, half, 1.0f);
    store_rows(a.dk + ((long)b * a.M + key) * (a.H * 64) + h * 64, ok, half, a.scale);
  }
}

// ---------------------------------------------------------------------------------------------------------
// Small-N form (round 5): q part and kv part of the backward for ONE (image, head) in ONE workgroup, for N, Mpad <= 128 -- the
// learn-NMS head's relation module (symbols/..._learn_nms.py:480-486: 100 ranked rois per (image, class), 640 pseudo-images at 8
// images per step).  The two-kernel form above spends its time on L2 round trips and on the fp32 S / dL maps
// ([640][16][100][128] x 4 B = 524 MB each: written by the q kernel, gathered column-wise by the kv kernel): 1.05 + 0.55 ms per
// step.  Here the 64-channel rows of K, VW, Q and dY of the (image, head) are staged in LDS once (72 KB), every query tile's wavefront reads its
// fragments from there, S and dL go to LDS as the bf16 values the kv products round them to anyway (70 KB) and are read back
// row-major by the key tiles' wavefronts after one barrier; only dL still goes to HBM (fp32, for the geometry backward).
// The transposed A operands of the dQ / dK / dVW products (K^T, Q^T, dY^T) are read from those row-major images on the fly (acc_pv_rm): no
// transposed copies are built or staged (the two-kernel form needs three transpose launches per module).  The q part's log G rows are read
// into registers before the first barrier: with one 4-wave workgroup per CU (140 KB of LDS) no other wavefront hides a load, so no global
// read sits inside the loops (first version, loads in the loops: 1.50 ms per learn-NMS launch = no gain over the two kernels' 1.05 + 0.55).
// Arithmetic and rounding points are those of the two-kernel form: same dq / dk / dvw / dlog bit for bit.
// ---------------------------------------------------------------------------------------------------------
constexpr int kSmK = 72, kSmT = 136;                                     // LDS row pitches (elements): 4 x odd words -> conflict-free fragment reads
constexpr int kSmallLdsBytes = 4 * 128 * kSmK * 2 + 2 * 128 * kSmT * 2;  // K | VW | Q | dY (row-major [row][64]) | S | dL = 143 360 B

// acc_pv with the A operand TRANSPOSED ON THE FLY from a row-major LDS image rm[x][kSmK] (x = key or query, 64 channels per row):
// o[d] += A^T[32 d + l31][x0 ...] * p, A^T[ch][x] = rm[x][ch].  Eight 2-byte LDS reads per fragment (lanes = consecutive channels of one row: 64
// contiguous bytes, conflict-free) instead of two 8-byte reads of a pre-transposed copy -- which costs a transpose launch and a second staged
// array per operand (K^T, Q^T, dY^T: 3 x 168 MB written and read back at the learn-NMS head, 0.28 ms of transposes per step).
__device__ __forceinline__ void acc_pv_rm(f32x16 (&o)[2], const unsigned short* rm, int x0, int l31, int half, const f32x16& p) {
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    bf16x8 pf;
    unsigned int* pw = (unsigned int*)&pf;
#pragma unroll
    for (int t = 0; t < 4; ++t) pw[t] = pack_bf16x2(p[8 * ks + 2 * t], p[8 * ks + 2 * t + 1]);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
      const unsigned short* col = rm + (x0 + 4 * half + 16 * ks) * kSmK + 32 * d + l31;
      bf16x8 af;
      unsigned int* aw = (unsigned int*)&af;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int xa = (t < 2 ? 0 : 8) + 2 * (t & 1);                  // k-slots 0-3 <-> x + 0..3, slots 4-7 <-> x + 8..11 (as acc_pv)
        aw[t] = (unsigned int)col[xa * kSmK] | ((unsigned int)col[(xa + 1) * kSmK] << 16);
      }
      o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, pf, o[d], 0, 0, 0);
    }
  }
}

// PACK: dQ / dK / dVW leave as bf16 straight into the [B][N][3 H 64] operand of the projection backward (dQ | dK | dVW column blocks: what
// relnet_relation_bwd_pack builds from the fp32 outputs otherwise -- 786 MB written and read back at the learn-NMS head); a.dq then points to
// that buffer, whose key blocks must be zero for rows >= M (the kernel never writes them).
template <bool PACK>
__global__ __launch_bounds__(256) void relation_attention_bwd_small_kernel(AttnBwdArgs a) {
  typedef unsigned short T;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
  T* sK = (T*)smem_s;                       // [128][kSmK]   K rows  (keys)
  T* sVW = sK + 128 * kSmK;                 // [128][kSmK]   VW rows (keys)
  T* sQ = sVW + 128 * kSmK;                 // [128][kSmK]   Q rows  (queries)
  T* sDY = sQ + 128 * kSmK;                 // [128][kSmK]   dY rows (queries)
  T* sP = sDY + 128 * kSmK;                 // [128][kSmT]   S  (bf16)
  T* sL = sP + 128 * kSmT;                  // [128][kSmT]   dL (bf16)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nqt = (a.N + 31) / 32, nkt = (a.M + 31) / 32;
  const int q = wave * 32 + l31, qc = q < a.N ? q : a.N - 1;
  const int npair = a.H * a.B;
  // Persistent over the (image, head) pairs with the NEXT pair's operands in flight: one 4-wave workgroup per CU (140 KB of LDS) means no other
  // wavefront hides a load -- counters of the non-persistent form: 58 % of the wave cycles parked in s_waitcnt / barriers, 20 % VALU.  So every
  // global read of a pair (16 staging chunks, the 16 float4 of this lane's log G row, its Y row) is issued while the PREVIOUS pair is being
  // computed and held in registers (one wave per SIMD: 512 registers).
  uint4 kK0, kK1, kK2, kK3, kV0, kV1, kV2, kV3, kQ0, kQ1, kQ2, kQ3, kD0, kD1, kD2, kD3;      // (named scalars: a loop-carried uint4[16] stayed in scratch)
  float4 bq[4][4];
  bf16x8 yv[4];
  // (a macro, not a lambda: with the arrays captured by reference the compiler kept kv[] in scratch -- and a scratch reload waits on vmcnt,
  //  i.e. for the very prefetch it belongs to)
#define RELNET_SMALL_LD(I)                                                                                                            \
  {                                                                                                                                   \
    const int c = tid + 256 * I, r = c >> 3, cc = c & 7;                                                                              \
    const int rk = r < a.M ? r : a.M - 1, rq = r < a.N ? r : a.N - 1;                                                                 \
    kK##I = *(const uint4*)(Kb + (long)rk * a.k_ld + cc * 8);                                                                         \
    kV##I = *(const uint4*)(VWb + (long)rk * a.vw_ld + cc * 8);                                                                       \
    kQ##I = *(const uint4*)(Qb + (long)rq * a.q_ld + cc * 8);                                                                         \
    kD##I = *(const uint4*)(DYb + (long)rq * a.dy_ld + cc * 8);                                                                       \
  }
#define RELNET_SMALL_ST(I)                                                                                                            \
  {                                                                                                                                   \
    const int c = tid + 256 * I, r = c >> 3, cc = c & 7;                                                                              \
    *(uint4*)(sK + r * kSmK + cc * 8) = kK##I;                                                                                        \
    *(uint4*)(sVW + r * kSmK + cc * 8) = kV##I;                                                                                       \
    *(uint4*)(sQ + r * kSmK + cc * 8) = kQ##I;                                                                                        \
    *(uint4*)(sDY + r * kSmK + cc * 8) = kD##I;                                                                                       \
  }
#define RELNET_SMALL_ISSUE(PAIR)                                                                                                      \
  {                                                                                                                                   \
    const int h_ = (PAIR) % a.H, b_ = (PAIR) / a.H;                                                                                   \
    const T* Kb = (const T*)a.k + (long)b_ * a.k_bs + h_ * 64;                                                                        \
    const T* VWb = (const T*)a.vw + (long)b_ * a.vw_bs + h_ * 64;                                                                     \
    const T* Qb = (const T*)a.q + (long)b_ * a.q_bs + h_ * 64;                                                                        \
    const T* DYb = (const T*)a.dy + (long)b_ * a.dy_bs + h_ * 64;                                                                     \
    RELNET_SMALL_LD(0) RELNET_SMALL_LD(1) RELNET_SMALL_LD(2) RELNET_SMALL_LD(3)                                                       \
    const float* Bq_ = a.bias + (long)b_ * a.bias_bs + ((long)h_ * a.N + qc) * a.Mpad;                                               \
    _Pragma("unroll") for (int kt = 0; kt < 4; ++kt)                                                                                  \
      _Pragma("unroll") for (int gq = 0; gq < 4; ++gq)                                                                                \
        bq[kt][gq] = (wave < nqt && kt < nkt) ? *(const float4*)(Bq_ + kt * 32 + 8 * gq + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f); \
    const T* Yr = (const T*)a.y + (long)b_ * a.y_bs + (long)qc * a.y_ld + h_ * 64;                                                    \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) yv[kk] = *(const bf16x8*)(Yr + 16 * kk + 8 * half);                              \
  }
  int pair = blockIdx.x;
  RELNET_SMALL_ISSUE(min(pair, npair - 1))
  for (; pair < npair; pair += gridDim.x) {
    const int h = pair % a.H, b = pair / a.H;
    // ---- this pair's operands: registers -> LDS (everyone left the previous pair's kv part at the barrier that ends the loop body)
    RELNET_SMALL_ST(0) RELNET_SMALL_ST(1) RELNET_SMALL_ST(2) RELNET_SMALL_ST(3)
    float4 bc[4][4];
    bf16x8 yc[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) bc[kt][gq] = bq[kt][gq];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) yc[kk] = yv[kk];
    __syncthreads();
    RELNET_SMALL_ISSUE(min(pair + (int)gridDim.x, npair - 1))           // in flight during this pair's arithmetic (unconditional: the last round re-reads
                                                                        // a valid pair -- a conditional refill keeps the arrays out of registers)
    const int Mb = a.key_count ? min(max(a.key_count[b], 1), a.M) : a.M;
    if (wave < nqt) {
      // ================= q part: this wavefront = 32 queries (relation_attention_bwd_q_kernel with the operands in LDS) =================
      float* Lq = (float*)a.dlog + (((long)b * a.H + h) * a.N + qc) * a.Mpad;
      Frag<T> qf, dyf;
      qf.load(sQ + q * kSmK, half);                                     // (row q < 128 of the staged image: rows >= N hold the last row, like qc)
      dyf.load(sDY + q * kSmK, half);
      float D = 0.f;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int d = 16 * kk + 8 * half + j;
          D += dyf.get(kk, j) * (bf2f((unsigned short)yc[kk][j]) - (a.bout ? a.bout[h * 64 + d] : 0.f));
        }
      D += __shfl_xor(D, 32);
      float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {                                  // pass 1: row maximum and normaliser
        if (kt >= nkt) break;
        const int key0 = kt * 32;
        f32x16 s = dot64<T>(sK + (key0 + l31) * kSmK, qf, half);
        float tmax = -INFINITY;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int kbase = key0 + 8 * gq + 4 * half;
          const float4 bv = bc[kt][gq];
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float v = bb[e] + a.scale * s[4 * gq + e];
            v = (kbase + e < Mb) ? v : -INFINITY;
            s[4 * gq + e] = v;
            tmax = fmaxf(tmax, v);
          }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) psum += sm_exp<T>(s[r] - m_new);
        l_run = l_run * sm_exp<T>(m_run - m_new) + psum;
        m_run = m_new;
      }
      const float inv = 1.0f / (l_run + __shfl_xor(l_run, 32));
      f32x16 o[2];
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {                                  // pass 2: S, dS, dL, dQ
        if (kt >= nkt) break;
        const int key0 = kt * 32;
        f32x16 s = dot64<T>(sK + (key0 + l31) * kSmK, qf, half);
        const f32x16 ds = dot64<T>(sVW + (key0 + l31) * kSmK, dyf, half);
        f32x16 dl;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int kbase = key0 + 8 * gq + 4 * half;
          const float4 bv = bc[kt][gq];
          const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
          float pv[4], lv[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * gq + e;
            const float v = bb[e] + a.scale * s[r];
            pv[e] = (kbase + e < Mb) ? sm_exp<T>(v - m_run) * inv : 0.f;
            lv[e] = pv[e] * (ds[r] - D);
            dl[r] = lv[e];
          }
          // S / dL for the key tiles' wavefronts: bf16 in LDS (exactly what acc_pv packs them to); dL for the geometry backward: fp32 in HBM
          *(uint2*)(sP + q * kSmT + kbase) = make_uint2(pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3]));
          *(uint2*)(sL + q * kSmT + kbase) = make_uint2(pack_bf16x2(lv[0], lv[1]), pack_bf16x2(lv[2], lv[3]));
          if (q < a.N) *(float4*)(Lq + kbase) = make_float4(lv[0], lv[1], lv[2], lv[3]);
        }
        acc_pv_rm(o, sK, key0, l31, half, dl);                          // dQ += dL K  (K^T read from the row-major image)
      }
      if (q < a.N) {
        if constexpr (PACK) store_rows_bf16((unsigned short*)a.dq + ((long)b * a.N + q) * (3 * a.H * 64) + h * 64, o, half, a.scale);
        else store_rows(a.dq + ((long)b * a.N + q) * (a.H * 64) + h * 64, o, half, a.scale);
      }
    }
    __syncthreads();
    if (wave < nkt) {
      // ================= kv part: this wavefront = 32 keys (relation_attention_bwd_kv_kernel with S / dL from LDS) =================
      const int key = wave * 32 + l31;
      f32x16 ov[2], ok[2];
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ov[d][r] = 0.f; ok[d][r] = 0.f; }
      for (int t = 0; t < nqt; ++t) {
        const int q0 = t * 32;
        f32x16 p, dl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int qq = q0 + 8 * (r >> 2) + 4 * half + (r & 3);
          const bool okq = qq < a.N;
          p[r] = okq ? bf2f(sP[qq * kSmT + key]) : 0.f;
          dl[r] = okq ? bf2f(sL[qq * kSmT + key]) : 0.f;
        }
        acc_pv_rm(ov, sDY, q0, l31, half, p);                           // dVW += S^T dY
        acc_pv_rm(ok, sQ, q0, l31, half, dl);                           // dK  += dL^T Q
      }
      if (key < a.M) {
        if constexpr (PACK) {
          unsigned short* row = (unsigned short*)a.dq + ((long)b * a.N + key) * (3 * a.H * 64) + h * 64;
          store_rows_bf16(row + a.H * 64, ok, half, a.scale);            // dK block
          store_rows_bf16(row + 2 * a.H * 64, ov, half, 1.0f);           // dVW block
        } else {
          store_rows(a.dvw + ((long)b * a.M + key) * (a.H * 64) + h * 64, ov, half, 1.0f);
          store_rows(a.dk + ((long)b * a.M + key) * (a.H * 64) + h * 64, ok, half, a.scale);
        }
      }
    }
    __syncthreads();                                                    // the LDS images are free for the next pair
  }
}

// ---------------------------------------------------------------------------------------------------------
struct GeomBwdArgs {
  const float* boxes; int box_stride, box_off;   // [B, N, box_stride]
  const float* bias;                              // [B][16][N][Mpad] fp32 log(max(G, 1e-6)) of this module
  const float* dlog;                              // [B][16][N][Mpad] fp32 dL
  float divisors[8];
  float* dwp;                                     // [16][64] (+=, atomics)
  float* dbp;                                     // [16]     (+=, atomics)
  int B, N, M, Mpad;
};

#pragma clang fp contract(off)
__device__ __forceinline__ float position_feature_1(float4 bi, float4 bj, int comp) {
  // relation.hip:position_features, one component
  const float wi = bi.z - bi.x + 1.f, hi = bi.w - bi.y + 1.f;
  const float wj = bj.z - bj.x + 1.f, hj = bj.w - bj.y + 1.f;
  const float cxi = 0.5f * (bi.x + bi.z), cyi = 0.5f * (bi.y + bi.w);
  const float cxj = 0.5f * (bj.x + bj.z), cyj = 0.5f * (bj.y + bj.w);
  float v;
  if (comp == 0) v = fmaxf(fabsf((cxi - cxj) / wi), 1e-3f);
  else if (comp == 1) v = fmaxf(fabsf((cyi - cyj) / hi), 1e-3f);
  else if (comp == 2) v = wi / wj;
  else v = hi / hj;
  return (float)log((double)v);
}
__device__ __forceinline__ void position_features(float4 bi, float4 bj, float (&p)[4]) {
  // identical to relation.hip:position_features (bit-exact fp32 sequence of SYM_REL:59-77, correctly rounded logs)
  const float wi = bi.z - bi.x + 1.f, hi = bi.w - bi.y + 1.f;
  const float wj = bj.z - bj.x + 1.f, hj = bj.w - bj.y + 1.f;
  const float cxi = 0.5f * (bi.x + bi.z), cyi = 0.5f * (bi.y + bi.w);
  const float cxj = 0.5f * (bj.x + bj.z), cyj = 0.5f * (bj.y + bj.w);
  const float dx = fmaxf(fabsf((cxi - cxj) / wi), 1e-3f);
  const float dy = fmaxf(fabsf((cyi - cyj) / hi), 1e-3f);
  p[0] = (float)log((double)dx);
  p[1] = (float)log((double)dy);
  p[2] = (float)log((double)(wi / wj));
  p[3] = (float)log((double)(hi / hj));
}
template <bool FAST>
__device__ __forceinline__ float embed_value(float p, int sc, float divisor) {
  const float arg = (100.0f * p) / divisor;
  if constexpr (FAST) return sc ? __cosf(arg) : __sinf(arg);      // v_cos / v_sin, like the forward's fp16-bias path
  else return sc ? cosf(arg) : sinf(arg);
}
#pragma clang fp contract(fast)

// one wavefront per (image, query i); MFMA k = pairs (two per instruction, one per half-wave).  The four fp64 logs of a
// pair are computed ONCE (lane L of a 64-pair block owns pair j0 + L, like the forward kernel's one-thread-per-pair) and
// handed to the lanes that need them with ds_bpermute; each lane then evaluates its two embedding columns.
template <bool FAST>
__global__ __launch_bounds__(256) void geometry_bias_bwd_kernel(GeomBwdArgs g) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int i = blockIdx.x * 4 + wave, b = blockIdx.y;
  if (i >= g.N) return;
  const float* bx = g.boxes + (long)b * g.N * g.box_stride + g.box_off;
  const float* pi = bx + (long)i * g.box_stride;
  const float4 bi = make_float4(pi[0], pi[1], pi[2], pi[3]);
  // this lane's two embedding columns: c = 32 blk + l31 -> component c >> 4, sin/cos (c >> 3) & 1, frequency c & 7
  const bool hi_comp = (l31 >> 4) != 0;          // column block 0: component 0 or 1; block 1: component 2 or 3
  const int sc = (l31 >> 3) & 1;
  const float div = g.divisors[l31 & 7];
  const int hh = l31 & 15;                       // head row supplied by this lane (rows >= 16 are zero)
  const float* Brow = g.bias + (((long)b * 16 + hh) * g.N + i) * g.Mpad;
  const float* Lrow = g.dlog + (((long)b * 16 + hh) * g.N + i) * g.Mpad;
  // clamped entries (relu(.) <= 1e-6 -> bias == ln 1e-6) carry no gradient.  The margin makes the test robust to how the
  // forward produced ln 1e-6 (libm logf, or log2 x ln 2 from the matrix-core kernel: last-bit differences around -13.8155)
  const float kLogFloor = logf(1e-6f) + 1e-3f;
  f32x16 c0, c1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
  float bsum = 0.f;
  for (int j0 = 0; j0 < g.M; j0 += 64) {
    // features of pair (i, j0 + lane)
    const int jl = j0 + lane < g.M ? j0 + lane : g.M - 1;
    const float* pj = bx + (long)jl * g.box_stride;
    float pf[4];
    position_features(bi, make_float4(pj[0], pj[1], pj[2], pj[3]), pf);
    const int nstep = min(32, (g.M - j0 + 1) >> 1);
    for (int t = 0; t < nstep; ++t) {
      const int src = 2 * t + half;              // lane that owns this half-wave's pair
      const int j = j0 + src;
      const bool okj = j < g.M;
      const float q0 = __shfl(pf[0], src), q1 = __shfl(pf[1], src), q2 = __shfl(pf[2], src), q3 = __shfl(pf[3], src);
      float dpre = 0.f;
      if (okj && l31 < 16) {
        const float lg = Brow[j];
        dpre = lg > kLogFloor ? Lrow[j] * (FAST ? __expf(-lg) : expf(-lg)) : 0.f;          // dL / G, zero on the clamped branch
      }
      bsum += dpre;
      const float e0 = okj ? embed_value<FAST>(hi_comp ? q1 : q0, sc, div) : 0.f;
      const float e1 = okj ? embed_value<FAST>(hi_comp ? q3 : q2, sc, div) : 0.f;
      c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(dpre, e0, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(dpre, e1, c1, 0, 0, 0);
    }
  }
  // rows h = (r & 3) + 8 (r >> 2) + 4 half < 16  <=>  r < 8 ; col = l31
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int hrow = (r & 3) + 8 * (r >> 2) + 4 * half;
    atomicAdd(g.dwp + hrow * 64 + l31, c0[r]);
    atomicAdd(g.dwp + hrow * 64 + 32 + l31, c1[r]);
  }
  bsum += __shfl_xor(bsum, 32);
  if (lane < 16) atomicAdd(g.dbp + lane, bsum);
}

// bf16 training path (FAST): the same product with the PAIRS on the contraction axis of the bf16 matrix cores
// (mfma_f32_32x32x16_bf16: 16 pairs per instruction instead of 2 on the fp32 MFMA), the operands vector-loaded (8 consecutive
// keys of one head row per lane) instead of one 4-byte load per (head, key), hardware log / sin / cos, and ONE atomic flush per
// workgroup after a grid-stride loop over the (image, query) rows (the per-wave flush of the kernel above put B N x 1024
// atomics on 1024 addresses).  A[h][pair] = dL / G (bf16), B[pair][f] = the embedding value (bf16, |.| <= 1), fp32 accumulate.
__global__ __launch_bounds__(256) void geometry_bias_bwd_mfma_kernel(GeomBwdArgs g, int total_q) {
  __shared__ __attribute__((aligned(16))) float sP[4][4][64];      // [wave][component][key of the 64-key block]
  __shared__ float sRed[3][16][65];
  __shared__ float sB[3][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int nw = gridDim.x * 4;
  const int comp0 = l31 >> 4, comp1 = 2 + (l31 >> 4), sc = (l31 >> 3) & 1;
  const float rate = 100.0f / g.divisors[l31 & 7];
  const int hh = l31 & 15;
  // clamped entries (relu(.) <= 1e-6 -> bias == ln 1e-6) carry no gradient.  The margin makes the test robust to how the
  // forward produced ln 1e-6 (libm logf, or log2 x ln 2 from the matrix-core kernel: last-bit differences around -13.8155)
  const float kLogFloor = logf(1e-6f) + 1e-3f;
  f32x16 c0, c1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { c0[r] = 0.f; c1[r] = 0.f; }
  float bsum = 0.f;
  for (int q = blockIdx.x * 4 + wave; q < total_q; q += nw) {
    const int b = q / g.N, i = q - b * g.N;
    const float* bx = g.boxes + (long)b * g.N * g.box_stride + g.box_off;
    const float* pi = bx + (long)i * g.box_stride;
    const float xi1 = pi[0], yi1 = pi[1], xi2 = pi[2], yi2 = pi[3];
    const float wi = xi2 - xi1 + 1.f, hi = yi2 - yi1 + 1.f, cxi = 0.5f * (xi1 + xi2), cyi = 0.5f * (yi1 + yi2);
    const float* Brow = g.bias + (((long)b * 16 + hh) * g.N + i) * g.Mpad;
    const float* Lrow = g.dlog + (((long)b * 16 + hh) * g.N + i) * g.Mpad;
    for (int j0 = 0; j0 < g.M; j0 += 64) {
      {   // the four position values of pair (i, j0 + lane), once per pair (SYM_REL:59-77, hardware log)
        const int jl = j0 + lane < g.M ? j0 + lane : g.M - 1;
        const float* pj = bx + (long)jl * g.box_stride;
        const float xj1 = pj[0], yj1 = pj[1], xj2 = pj[2], yj2 = pj[3];
        const float wj = xj2 - xj1 + 1.f, hj = yj2 - yj1 + 1.f, cxj = 0.5f * (xj1 + xj2), cyj = 0.5f * (yj1 + yj2);
        sP[wave][0][lane] = __logf(fmaxf(fabsf((cxi - cxj) / wi), 1e-3f));
        sP[wave][1][lane] = __logf(fmaxf(fabsf((cyi - cyj) / hi), 1e-3f));
        sP[wave][2][lane] = __logf(wi / wj);
        sP[wave][3][lane] = __logf(hi / hj);
      }
      const int nstep = (min(64, g.M - j0) + 15) >> 4;
      for (int kk = 0; kk < nstep; ++kk) {
        const int pb = 16 * kk + 8 * half, jb = j0 + pb;
        float dpre[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) dpre[t] = 0.f;
        if (l31 < 16) {
          const float4 l0 = *(const float4*)(Brow + jb), l1 = *(const float4*)(Brow + jb + 4);
          const float4 d0 = *(const float4*)(Lrow + jb), d1 = *(const float4*)(Lrow + jb + 4);
          const float lg[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
          const float dl[8] = {d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
#pragma unroll
          for (int t = 0; t < 8; ++t)
            dpre[t] = (jb + t < g.M && lg[t] > kLogFloor) ? dl[t] * __expf(-lg[t]) : 0.f;      // dL / G, zero on the clamped branch
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) bsum += dpre[t];
        const float4 p00 = *(const float4*)&sP[wave][comp0][pb], p01 = *(const float4*)&sP[wave][comp0][pb + 4];
        const float4 p10 = *(const float4*)&sP[wave][comp1][pb], p11 = *(const float4*)&sP[wave][comp1][pb + 4];
        const float q0[8] = {p00.x, p00.y, p00.z, p00.w, p01.x, p01.y, p01.z, p01.w};
        const float q1[8] = {p10.x, p10.y, p10.z, p10.w, p11.x, p11.y, p11.z, p11.w};
        union { bf16x8 f; unsigned int u[4]; } fa, fb0, fb1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const bool ok0 = jb + 2 * t < g.M, ok1 = jb + 2 * t + 1 < g.M;
          const float a0 = q0[2 * t] * rate, a1 = q0[2 * t + 1] * rate, b0 = q1[2 * t] * rate, b1 = q1[2 * t + 1] * rate;
          const float e00 = ok0 ? (sc ? __cosf(a0) : __sinf(a0)) : 0.f, e01 = ok1 ? (sc ? __cosf(a1) : __sinf(a1)) : 0.f;
          const float e10 = ok0 ? (sc ? __cosf(b0) : __sinf(b0)) : 0.f, e11 = ok1 ? (sc ? __cosf(b1) : __sinf(b1)) : 0.f;
          fa.u[t] = pack_bf16x2(dpre[2 * t], dpre[2 * t + 1]);
          fb0.u[t] = pack_bf16x2(e00, e01);
          fb1.u[t] = pack_bf16x2(e10, e11);
        }
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.f, fb0.f, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa.f, fb1.f, c1, 0, 0, 0);
      }
    }
  }
  // rows h = (r & 3) + 8 (r >> 2) + 4 half < 16  <=>  r < 8 ; col = l31.  Waves 1-3 hand their tile to wave 0.
  bsum += __shfl_xor(bsum, 32);
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int hrow = (r & 3) + 8 * (r >> 2) + 4 * half;
      sRed[wave - 1][hrow][l31] = c0[r];
      sRed[wave - 1][hrow][32 + l31] = c1[r];
    }
    if (lane < 16) sB[wave - 1][lane] = bsum;
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int hrow = (r & 3) + 8 * (r >> 2) + 4 * half;
      float v0 = c0[r], v1 = c1[r];
#pragma unroll
      for (int w = 0; w < 3; ++w) { v0 += sRed[w][hrow][l31]; v1 += sRed[w][hrow][32 + l31]; }
      atomicAdd(g.dwp + hrow * 64 + l31, v0);
      atomicAdd(g.dwp + hrow * 64 + 32 + l31, v1);
    }
    if (lane < 16) atomicAdd(g.dbp + lane, bsum + sB[0][lane] + sB[1][lane] + sB[2][lane]);
  }
}

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_transpose_2d(const void* in, long in_ld, long in_bs, void* out, long out_ld, long out_bs,
                                   int rows, int cols, int batch, int dtype, void* stream) {
  RELNET_REQUIRE(in && out, "relnet_transpose_2d: null operand");
  RELNET_REQUIRE(rows > 0 && cols > 0 && batch > 0 && in_ld >= cols && out_ld >= rows, "relnet_transpose_2d: bad shape");
  dim3 grid((cols + 63) / 64, (rows + 63) / 64, batch);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RELNET_F32) transpose_2d_kernel<float><<<grid, 256, 0, s>>>((const float*)in, in_ld, in_bs, (float*)out, out_ld, out_bs, rows, cols);
  else if (dtype == RELNET_BF16) {
    // vector path: whole 8-element groups on both sides (the destination rows are padded by the caller, so writing a
    // group that extends past `rows` up to out_ld is allowed only when it stays inside the row: rows rounded to 8 <= out_ld)
    const bool vec = cols % 8 == 0 && in_ld % 8 == 0 && in_bs % 8 == 0 && out_ld % 8 == 0 && out_bs % 8 == 0 &&
                     ((rows + 7) / 8) * 8 <= out_ld && (((uintptr_t)in | (uintptr_t)out) & 15) == 0;
    if (vec) transpose_2d_bf16_vec_kernel<<<grid, 256, 0, s>>>((const unsigned short*)in, in_ld, in_bs, (unsigned short*)out, out_ld, out_bs, rows, cols);
    else transpose_2d_kernel<unsigned short><<<grid, 256, 0, s>>>((const unsigned short*)in, in_ld, in_bs, (unsigned short*)out, out_ld, out_bs, rows, cols);
  }
  else RELNET_REQUIRE(false, "relnet_transpose_2d: unknown dtype %d", dtype);
  return check_launch("relnet_transpose_2d");
}

extern "C" int relnet_relation_attention_bwd_kc(
    const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs, const void* kt, long kt_ld, long kt_bs,
    const void* vw, long vw_ld, long vw_bs, const float* bias, long bias_bs, const void* dy, long dy_ld, long dy_bs,
    const void* y, long y_ld, long y_bs, const float* bout, const void* qt, long qt_ld, long qt_bs, const void* dyt,
    long dyt_ld, long dyt_bs, float* prob, float* dlog, float* dq, float* dk, float* dvw, int B, int H, int N, int M,
    int Mpad, int Npad, float scale, int dtype, const int* key_count, void* stream) {
  RELNET_REQUIRE(q && k && vw && bias && dy && y && dlog && dq, "relnet_relation_attention_bwd: null operand");
  RELNET_REQUIRE((kt && qt && dyt) || !prob, "relnet_relation_attention_bwd: the transposed operands kt / qt / dyt may be NULL only for the small-N form (prob NULL)");
  // dk == dvw == NULL (with prob == NULL): dq is the bf16 [B][N][3 H 64] operand of the projection backward and receives (dQ | dK | dVW)
  RELNET_REQUIRE((dk && dvw) || (!dk && !dvw && !prob), "relnet_relation_attention_bwd: dk and dvw are given together, or both NULL with prob NULL (packed bf16 output in dq)");
  // prob == NULL selects the one-workgroup-per-(image, head) form (relation_attention_bwd_small_kernel): bf16, N and Mpad <= 128
  RELNET_REQUIRE(prob || (dtype == RELNET_BF16 && N <= 128 && Mpad <= 128),
                 "relnet_relation_attention_bwd: prob may be NULL only for bf16 operands with N, Mpad <= 128 (N=%d Mpad=%d dtype=%d)", N, Mpad, dtype);
  RELNET_REQUIRE(B > 0 && H > 0 && N > 0 && M > 0 && M <= N && Mpad >= M && Mpad % 32 == 0 && Npad >= N && Npad % 32 == 0,
                 "relnet_relation_attention_bwd: bad shape (N=%d M=%d Mpad=%d Npad=%d)", N, M, Mpad, Npad);
  RELNET_REQUIRE(!prob || (kt_ld >= Mpad && qt_ld >= Npad && dyt_ld >= Npad), "relnet_relation_attention_bwd: transposed operands must be padded");
  AttnBwdArgs a;
  a.q = q; a.q_ld = q_ld; a.q_bs = q_bs; a.k = k; a.k_ld = k_ld; a.k_bs = k_bs; a.kt = kt; a.kt_ld = kt_ld; a.kt_bs = kt_bs;
  a.vw = vw; a.vw_ld = vw_ld; a.vw_bs = vw_bs; a.bias = bias; a.bias_bs = bias_bs; a.dy = dy; a.dy_ld = dy_ld; a.dy_bs = dy_bs;
  a.y = y; a.y_ld = y_ld; a.y_bs = y_bs; a.bout = bout; a.qt = qt; a.qt_ld = qt_ld; a.qt_bs = qt_bs;
  a.dyt = dyt; a.dyt_ld = dyt_ld; a.dyt_bs = dyt_bs; a.prob = prob; a.dlog = dlog; a.dq = dq; a.dk = dk; a.dvw = dvw;
  a.B = B; a.H = H; a.N = N; a.M = M; a.Mpad = Mpad; a.scale = scale; a.key_count = key_count;
  hipStream_t s = (hipStream_t)stream;
  dim3 gq((unsigned)(((N + 31) / 32 + 3) / 4), H, B), gk((unsigned)(((M + 31) / 32 + 3) / 4), H, B);
  if (!prob) {
    RELNET_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vw_ld % 8 == 0 && dy_ld % 8 == 0 && y_ld % 8 == 0 && q_bs % 8 == 0 && k_bs % 8 == 0 && vw_bs % 8 == 0 &&
                   dy_bs % 8 == 0 && y_bs % 8 == 0 && bias_bs % 4 == 0 &&
                   (((uintptr_t)q | (uintptr_t)dy | (uintptr_t)k | (uintptr_t)vw | (uintptr_t)y | (uintptr_t)bias) & 15) == 0,
                   "relnet_relation_attention_bwd(bf16, small-N form): rows of every operand (q, k, vw, dy, y, bias) must be 16-byte aligned");
    static relnet::PerDeviceOnce attr_once;
    if (attr_once.first()) {
      const hipError_t e1 = hipFuncSetAttribute((const void*)relation_attention_bwd_small_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      const hipError_t e2 = hipFuncSetAttribute((const void*)relation_attention_bwd_small_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      RELNET_REQUIRE(e1 == hipSuccess && e2 == hipSuccess, "relnet_relation_attention_bwd: the small-N kernel needs 160 KB of LDS per workgroup (hipFuncSetAttribute: %s)",
                     hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    }
    const long cus = relnet::device_cu_count();
    const unsigned grid = (unsigned)((long)H * B < cus ? (long)H * B : cus);      // persistent: one workgroup per CU walks the (image, head) pairs
    if (dk) relation_attention_bwd_small_kernel<false><<<grid, 256, relnet::kSmallLdsBytes, s>>>(a);
    else relation_attention_bwd_small_kernel<true><<<grid, 256, relnet::kSmallLdsBytes, s>>>(a);
    return check_launch("relnet_relation_attention_bwd");
  }
  if (dtype == RELNET_F32) {
    RELNET_REQUIRE(q_ld % 4 == 0 && k_ld % 4 == 0 && kt_ld % 4 == 0 && vw_ld % 4 == 0 && dy_ld % 4 == 0 && qt_ld % 4 == 0 && dyt_ld % 4 == 0,
                   "relnet_relation_attention_bwd(f32): rows must be 16-byte aligned");
    relation_attention_bwd_q_kernel<float><<<gq, 256, 0, s>>>(a);
    relation_attention_bwd_kv_kernel<float><<<gk, 256, 0, s>>>(a, Npad);
  } else if (dtype == RELNET_BF16) {
    RELNET_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && kt_ld % 4 == 0 && vw_ld % 8 == 0 && dy_ld % 8 == 0 && y_ld % 8 == 0 && qt_ld % 4 == 0 && dyt_ld % 4 == 0,
                   "relnet_relation_attention_bwd(bf16): rows must be 16-byte (8-byte for the transposed operands) aligned");
    relation_attention_bwd_q_kernel<unsigned short><<<gq, 256, 0, s>>>(a);
    relation_attention_bwd_kv_kernel<unsigned short><<<gk, 256, 0, s>>>(a, Npad);
  } else {
    RELNET_REQUIRE(false, "relnet_relation_attention_bwd: unknown dtype %d", dtype);
  }
  return check_launch("relnet_relation_attention_bwd");
}

extern "C" int relnet_relation_attention_bwd(
    const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs, const void* kt, long kt_ld, long kt_bs,
    const void* vw, long vw_ld, long vw_bs, const float* bias, long bias_bs, const void* dy, long dy_ld, long dy_bs,
    const void* y, long y_ld, long y_bs, const float* bout, const void* qt, long qt_ld, long qt_bs, const void* dyt,
    long dyt_ld, long dyt_bs, float* prob, float* dlog, float* dq, float* dk, float* dvw, int B, int H, int N, int M,
    int Mpad, int Npad, float scale, int dtype, void* stream) {
  return relnet_relation_attention_bwd_kc(q, q_ld, q_bs, k, k_ld, k_bs, kt, kt_ld, kt_bs, vw, vw_ld, vw_bs, bias, bias_bs, dy, dy_ld,
                                          dy_bs, y, y_ld, y_bs, bout, qt, qt_ld, qt_bs, dyt, dyt_ld, dyt_bs, prob, dlog, dq, dk, dvw,
                                          B, H, N, M, Mpad, Npad, scale, dtype, nullptr, stream);
}

extern "C" int relnet_geometry_bias_bwd(const float* boxes, int box_stride, int box_off, const float* bias,
                                        const float* dlog, const float* divisors8, float* dwp, float* dbp, int B,
                                        int N, int M, int Mpad, int fast_math, void* stream) {
  RELNET_REQUIRE(boxes && bias && dlog && divisors8 && dwp && dbp, "relnet_geometry_bias_bwd: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && M > 0 && Mpad >= M, "relnet_geometry_bias_bwd: bad shape");
  GeomBwdArgs g;
  g.boxes = boxes; g.box_stride = box_stride; g.box_off = box_off; g.bias = bias; g.dlog = dlog;
  for (int k = 0; k < 8; ++k) g.divisors[k] = divisors8[k];
  g.dwp = dwp; g.dbp = dbp; g.B = B; g.N = N; g.M = M; g.Mpad = Mpad;
  dim3 grid((unsigned)((N + 3) / 4), B);
  if (fast_math == 1 && Mpad % 32 == 0 && (((uintptr_t)bias | (uintptr_t)dlog) & 15) == 0) {
    const long total_q = (long)B * N;             // grid-stride over the (image, query) rows: <= 768 workgroups
    const long blocks = (total_q + 3) / 4;
    geometry_bias_bwd_mfma_kernel<<<(unsigned)(blocks < 768 ? blocks : 768), 256, 0, (hipStream_t)stream>>>(g, (int)total_q);
  } else if (fast_math) geometry_bias_bwd_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  else geometry_bias_bwd_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_geometry_bias_bwd");
}
