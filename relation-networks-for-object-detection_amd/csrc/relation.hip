// Object-relation module kernels (reference: relation_rcnn/symbols/
// resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py = SYM_REL).
//
//  geometry_bias_kernel   SYM_REL:46-83 (extract_position_matrix) + :29-44
//                         (extract_position_embedding) + :109-116 (pair_pos_fc1 + ReLU) +
//                         the log(max(.,1e-6)) of :139, fused: the [N,M,64] embedding
//                         (23 MB / image in the reference) never exists in HBM; the same
//                         sin/cos values feed every relation module that shares the boxes.
//  relation_attention_kernel  SYM_REL:132-150: logits = bias + QK^T/8, softmax over keys,
//                         value aggregation and the grouped 1x1 `linear_out`.
//                         linear_out is re-associated: Y_h = S_h (F_K Wout_h^T); the
//                         [M,64] product F_K Wout_h^T ("VW") comes from the GEMM kernel as
//                         VW^T[h*64+dv][key], so the kernel is a flash-style attention
//                         with d_k = d_v = 64 (executed FLOPs 4.4x below the graph as
//                         written; DESIGN.md reports both).
#include "common.h"
#include <type_traits>
#include <hip/hip_fp16.h>
#include <stdlib.h>

namespace relnet {

// ---------------------------------------------------------------------------------------
// geometry bias: one thread per (query i, key j) pair, all heads x modules.
// ---------------------------------------------------------------------------------------
struct GeomArgs {
  const float* boxes;      // [B, N, box_stride] (x1,y1,x2,y2 at +box_off)
  int box_stride, box_off;
  const float* wp;         // [64, NMOD*FC]  pair_pos_fc1 weights, embedding-index major
  const float* bp;         // [NMOD, FC]
  float divisors[8];       // wave_length^(k/8), fp32 (host computes them like the graph)
  float c2[8];             // ln2 * 100 / (2 pi divisors[k]): log2-domain position value -> revolutions (MFMA kernel)
  void* bias;              // [NMOD, B, FC, N, Mpad]   log(max(relu(E Wp^T + bp), 1e-6)) float, or log2(.) half
  float* pos_mat;          // optional [B, N, M, 4]
  float* pos_emb;          // optional [B, N, M, 64]
  int B, N, M, Mpad, nmod;
  int out_f32;             // geometry_bias_mfma_kernel: store float32 ln(.) instead of fp16 log2(.) (training backward's recompute)
};

#pragma clang fp contract(off)
__device__ __forceinline__ void position_features(float4 bi, float4 bj, float (&p)[4]) {
  // bit-for-bit the fp32 op sequence of SYM_REL:59-77; log is correctly rounded (fp64
  // evaluation) because a 1-ulp log difference is amplified x100 before sin/cos.
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

// EXACT (float output only): the float32 PARITY form -- sin / cos / log correctly rounded (float64 evaluation, one rounding)
// and the 64 -> 16 pair_pos_fc1 product accumulated in float64 and rounded once, i.e. the arithmetic of oracle/relation.py
// (`cr`, `_mm`): the logits `weighted_aff` (SYM_REL:139) then agree with the oracle to float32 rounding for EVERY pair,
// also where log(max(G, 1e-6)) amplifies G's last bits.  !EXACT keeps libm float32 sincosf / logf and float32 FMAs (the
// training backward's recompute), fp16 output = hardware v_sin / v_cos / v_log (throughput path without the MFMA kernel).
template <int FC, int NMOD, typename TB, bool EXACT = false>
__global__ __launch_bounds__(256) void geometry_bias_kernel(GeomArgs g) {
  constexpr bool kFast = sizeof(TB) == 2;    // fp16 bias = bf16 throughput path: hardware sin/cos
  const long pair = (long)blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y;
  if (pair >= (long)g.N * g.M) return;
  const int i = (int)(pair / g.M), j = (int)(pair % g.M);
  const float* bx = g.boxes + (long)b * g.N * g.box_stride + g.box_off;
  const float* pi = bx + (long)i * g.box_stride;
  const float* pj = bx + (long)j * g.box_stride;
  const float4 bi = make_float4(pi[0], pi[1], pi[2], pi[3]);
  const float4 bj = make_float4(pj[0], pj[1], pj[2], pj[3]);
  float p[4];
  position_features(bi, bj, p);
  if (g.pos_mat) *(float4*)(g.pos_mat + (((long)b * g.N + i) * g.M + j) * 4) = make_float4(p[0], p[1], p[2], p[3]);

  constexpr int NO = NMOD * FC;                  // outputs per pair
  typedef typename std::conditional<EXACT, double, float>::type ACC;
  ACC acc[NO];
#pragma unroll
  for (int o = 0; o < NO; ++o) acc[o] = (ACC)((const float __attribute__((address_space(4))) *)(unsigned long long)g.bp)[o];

#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    const float pc = c == 0 ? p[0] : c == 1 ? p[1] : c == 2 ? p[2] : p[3];
    const float x100 = 100.0f * pc;
    float e[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float arg = x100 / g.divisors[k];
      if constexpr (kFast) {       // v_sin/v_cos (|arg| <= 700 rad is inside their range); abs err ~1e-5
        e[k] = __sinf(arg); e[8 + k] = __cosf(arg);
      } else if constexpr (EXACT) {
        double sn, cs;
        sincos((double)arg, &sn, &cs);
        e[k] = (float)sn; e[8 + k] = (float)cs;
      } else {
        sincosf(arg, &e[k], &e[8 + k]);
      }
    }
    if (g.pos_emb) {
      float* pe = g.pos_emb + (((long)b * g.N + i) * g.M + j) * 64 + c * 16;
#pragma unroll
      for (int k = 0; k < 16; ++k) pe[k] = e[k];
    }
    // wp_t is [64][NO] (embedding index major): the NO weights of one embedding element
    // are contiguous and wave-uniform -> s_load_dwordx16 + v_fmac with an SGPR operand.
    // constant address space => s_load_dwordx8/16 into SGPRs (a plain pointer gives per-lane
    // global_load_dwordx4 of the same address: hipcc cannot prove the weights are not aliased
    // by the bias stores)
    typedef const float __attribute__((address_space(4))) * cfloat_p;
    cfloat_p w = (cfloat_p)(unsigned long long)(g.wp + (long)(c * 16) * NO);
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int o = 0; o < NO; ++o) {
        if constexpr (EXACT) acc[o] += (double)e[k] * (double)w[k * NO + o];      // exact product, float64 sum
        else acc[o] = fmaf(e[k], w[k * NO + o], acc[o]);
      }
  }
#pragma unroll
  for (int m = 0; m < NMOD; ++m)
#pragma unroll
    for (int h = 0; h < FC; ++h) {
      const float gw = fmaxf(fmaxf((float)acc[m * FC + h], 0.f), 1e-6f);
      // fp16 bias feeds the exp2-based LDS attention kernel: store log2(G) (v_log_f32 is log2)
      ((TB*)g.bias)[((((long)m * g.B + b) * FC + h) * g.N + i) * g.Mpad + j] =
          (TB)(kFast ? __log2f(gw) : EXACT ? (float)log((double)gw) : logf(gw));
    }
}
#pragma clang fp contract(fast)

// ---------------------------------------------------------------------------------------
// geometry bias, throughput form (fp16 log2 G for the bf16 attention kernel): the 64 -> 16 pair_pos_fc1 product of
// BOTH relation modules is one 32 x 32 x 64 MFMA problem per 32 (query, key) pairs -- rows = (module, head), columns =
// pairs -- instead of 2048 scalar FMAs per pair (the VALU kernel above spends ~80 % of its 324 us at B = 54 there).
// One wavefront = one query x 64 keys per step (two 32-pair sets, keys 2 l31 and 2 l31 + 1, so that a lane packs two
// adjacent keys into one 4-byte store and a (module, head) row leaves as 128 contiguous bytes).  Lane (l31, half)
// evaluates, per coordinate c = k-step, the 8 sines (half 0) or cosines (half 1) of its pair: feature
// f = 16 c + 8 half + t (SYM_REL:29-44) is exactly the B-operand slot of mfma_f32_32x32x16_f16.  Features and weights
// are fp16 in the product (|feature| <= 1), accumulation fp32; hardware log2 / sin as in the kFast VALU path.
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// two fp32 -> packed fp16, round to nearest even: the vector fptrunc selects gfx950's v_cvt_pk_f16_f32
// (__builtin_amdgcn_cvt_pkrtz truncates; inline asm would hide the trans -> VALU forwarding hazard of its v_sin inputs
// from the compiler's hazard recogniser -- measured: wrong results)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_f16x2_rn(float lo, float hi) {
  const f32x2 v = {lo, hi};
  return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, f16x2));
}

__global__ __launch_bounds__(256) void geometry_bias_mfma_kernel(GeomArgs g) {
  extern __shared__ __attribute__((aligned(16))) float sKey[];   // [nblk * 64][4]: cx, cy, log2 w, log2 h of the keys
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int b = blockIdx.y;
  const int NO = g.nmod * 16;
  const int nblk = (g.M + 63) / 64;
  const float* bx = g.boxes + (long)b * g.N * g.box_stride + g.box_off;
  for (int j = tid; j < nblk * 64; j += 256) {
    const float* pj = bx + (long)(j < g.M ? j : g.M - 1) * g.box_stride;
    *(float4*)(sKey + 4 * j) = make_float4(0.5f * (pj[0] + pj[2]), 0.5f * (pj[1] + pj[3]),
                                            __builtin_amdgcn_logf(pj[2] - pj[0] + 1.f), __builtin_amdgcn_logf(pj[3] - pj[1] + 1.f));
  }
  // pair_pos_fc1 fragments (A operand): row = l31 = module * 16 + head, k = 16 kk + 8 half + t; wp is [64][NO]
  f16x8 wf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk)
#pragma unroll
    for (int t = 0; t < 8; ++t) wf[kk][t] = l31 < NO ? (_Float16)g.wp[(16 * kk + 8 * half + t) * NO + l31] : (_Float16)0.f;
  f32x16 bpv;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = 8 * (r >> 2) + 4 * half + (r & 3);
    bpv[r] = row < NO ? g.bp[row] : 1.f;
  }
  const float phase = half ? 0.25f : 0.f;                 // cos x = sin(x + 1/4 revolution)
  __syncthreads();
  const int i = blockIdx.x * 4 + wave;
  if (i >= g.N) return;
  const float* pi = bx + (long)i * g.box_stride;
  const float wi = pi[2] - pi[0] + 1.f, hi = pi[3] - pi[1] + 1.f;
  const float cxi = 0.5f * (pi[0] + pi[2]), cyi = 0.5f * (pi[1] + pi[3]);
  const float iwi = 1.0f / wi, ihi = 1.0f / hi, l2wi = __builtin_amdgcn_logf(wi), l2hi = __builtin_amdgcn_logf(hi);
  __half* out = (__half*)g.bias;
  for (int jb = 0; jb < nblk; ++jb) {
    f32x16 acc[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const float4 kb = *(const float4*)(sKey + 4 * (jb * 64 + 2 * l31 + ps));
      float p[4];
      p[0] = __builtin_amdgcn_logf(fmaxf(fabsf(cxi - kb.x) * iwi, 1e-3f));     // SYM_REL:59-66 (log2; ln 2 folded into c2)
      p[1] = __builtin_amdgcn_logf(fmaxf(fabsf(cyi - kb.y) * ihi, 1e-3f));
      p[2] = l2wi - kb.z;                                                       // log2(w_i / w_j), SYM_REL:67-70
      p[3] = l2hi - kb.w;
      acc[ps] = bpv;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        u32x4 w;
#pragma unroll
        for (int t = 0; t < 8; t += 2)
          w[t >> 1] = pack_f16x2_rn(__builtin_amdgcn_sinf(fmaf(p[kk], g.c2[t], phase)), __builtin_amdgcn_sinf(fmaf(p[kk], g.c2[t + 1], phase)));
        acc[ps] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk], __builtin_bit_cast(f16x8, w), acc[ps], 0, 0, 0);
      }
    }
    const int j2 = jb * 64 + 2 * l31;
    if (j2 >= g.Mpad) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = 8 * (r >> 2) + 4 * half + (r & 3);
      if (row >= NO) continue;
      const int m = row >> 4, h = row & 15;
      const float v0 = __builtin_amdgcn_logf(fmaxf(acc[0][r], 1e-6f));        // relu, clamp (SYM_REL:116,139), log2
      const float v1 = __builtin_amdgcn_logf(fmaxf(acc[1][r], 1e-6f));
      const long o = ((((long)m * g.B + b) * 16 + h) * g.N + i) * g.Mpad + j2;
      if (g.out_f32) *(float2*)((float*)g.bias + o) = make_float2(v0 * 0.69314718055994530942f, v1 * 0.69314718055994530942f);
      else *(__half2*)(out + o) = __floats2half2_rn(v0, v1);
    }
  }
}

// ---------------------------------------------------------------------------------------
// relation attention.  One wave = 32 queries of one (image, head); key tiles of 32.
// S^T = K Q^T is computed "swapped" so every lane owns one query column: the row max /
// row sum are in-lane plus one lane^32 exchange, and the per-query rescale is a per-lane
// scalar.  P^T feeds the second MFMA directly from the accumulator registers: the
// contraction (key) index is permuted identically in P^T and in the VW^T fragment loads,
// so no cross-lane shuffle or LDS round trip is needed between the two products.
// ---------------------------------------------------------------------------------------
struct AttnArgs {
  const void* q; long q_ld, q_bs;        // [B][N][.. h*64+d ..]
  const void* k; long k_ld, k_bs;        // [B][M][.. h*64+d ..]
  const void* vwt; long vwt_ld, vwt_bs;  // [B][H*64][Mpad]   (VW^T, keys contiguous)
  const void* bias; long bias_bs;        // [B][H][N][Mpad] fp32 (or fp16 for the LDS kernel)
  const float* bout;                     // [H*64] linear_out bias or nullptr
  const void* resid; long resid_ld, resid_bs;   // optional residual (same dtype as out)
  void* out; long out_ld, out_bs;        // Y = attention output (nullptr to skip)
  void* out_act; long act_ld, act_bs;    // relu(resid + Y)      (nullptr to skip)
  float* logits;                         // optional [B][N][H][M] fp32 (weighted_aff)
  int B, H, N, M, Mpad;
  float scale;
  const int* key_count;                  // optional [B]: only the first key_count[b] (<= M) keys of image b exist; the others
                                         // are masked like the columns past M (padding rows of a fixed-size roi buffer)
};

template <typename T, typename TOUT>
__global__ __launch_bounds__(256) void relation_attention_kernel(AttnArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int qt = blockIdx.x * 4 + wave;
  const int h = blockIdx.y, b = blockIdx.z;
  if (qt * 32 >= a.N) return;
  const int q = qt * 32 + l31;
  const int qc = q < a.N ? q : a.N - 1;                 // clamped row for loads
  constexpr bool kBF = sizeof(T) == 2;

  const T* Q = (const T*)a.q + (long)b * a.q_bs + (long)qc * a.q_ld + h * 64;
  const T* Kb = (const T*)a.k + (long)b * a.k_bs + h * 64;
  const T* Vb = (const T*)a.vwt + (long)b * a.vwt_bs + (long)(h * 64) * a.vwt_ld;
  const float* Bq = (const float*)a.bias + (long)b * a.bias_bs + ((long)h * a.N + qc) * a.Mpad;

  // Q fragments (B operand of S^T = K Q^T): 64 d-values per query.
  bf16x8 qf[4];
  float qs[32];
  if constexpr (kBF) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(Q + 16 * kk + 8 * half);
  } else {
#pragma unroll
    for (int s = 0; s < 32; s += 4) {
      const float4 v = *(const float4*)(Q + half * 32 + s);
      qs[s] = v.x; qs[s + 1] = v.y; qs[s + 2] = v.z; qs[s + 3] = v.w;
    }
  }

  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  const int Mb = a.key_count ? min(max(a.key_count[b], 1), a.M) : a.M;      // keys of THIS image
  const int nkt = (a.M + 31) / 32;
  for (int kt = 0; kt < nkt; ++kt) {
    const int key0 = kt * 32;
    int kr = key0 + l31; kr = kr < a.M ? kr : a.M - 1;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    if constexpr (kBF) {
      const T* Kr = Kb + (long)kr * a.k_ld;
      bf16x8 kf[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[kk] = *(const bf16x8*)(Kr + 16 * kk + 8 * half);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s, 0, 0, 0);
    } else {
      const float* Kr = (const float*)Kb + (long)kr * a.k_ld + half * 32;
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 v = *(const float4*)(Kr + 4 * c4);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, qs[4 * c4 + 0], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, qs[4 * c4 + 1], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, qs[4 * c4 + 2], s, 0, 0, 0);
        s = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, qs[4 * c4 + 3], s, 0, 0, 0);
      }
    }
    // logits for this lane's query: keys key0 + 8g + 4*half + (0..3), g = r >> 2
    float tmax = -INFINITY;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int kbase = key0 + 8 * gq + 4 * half;
      const float4 bv = *(const float4*)(Bq + kbase);
      const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * gq + e;
        float v = a.scale * s[r];
        v = bb[e] + v;                                   // weighted_aff (SYM_REL:139)
        v = (kbase + e < Mb) ? v : -INFINITY;
        s[r] = v;
        tmax = fmaxf(tmax, v);
      }
      if (a.logits && q < a.N) {
        float* lp = a.logits + (((long)b * a.N + q) * a.H + h) * a.M + kbase;
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kbase + e < a.M) lp[e] = s[4 * gq + e];
      }
    }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    const float m_new = fmaxf(m_run, tmax);
    const float alpha = kBF ? __expf(m_run - m_new) : expf(m_run - m_new);   // first tile: exp(-inf) = 0
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = kBF ? __expf(s[r] - m_new) : expf(s[r] - m_new);
      s[r] = pv;
      psum += pv;
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

    // O^T[dv][q] += VW^T[dv][key] P^T[key][q]; key slot t of k-step ks is
    // key0 + 16 ks + 8 (t >> 2) + 4 half + (t & 3) for BOTH operands.
    if constexpr (kBF) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8 pf;
        unsigned int* pw = (unsigned int*)&pf;
#pragma unroll
        for (int t = 0; t < 4; ++t) pw[t] = pack_bf16x2(s[8 * ks + 2 * t], s[8 * ks + 2 * t + 1]);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
          const T* Vr = Vb + (long)(32 * d + l31) * a.vwt_ld + key0 + 16 * ks + 4 * half;
          bf16x8 vf;
          *(uint2*)&vf = *(const uint2*)(Vr);
          *((uint2*)&vf + 1) = *(const uint2*)(Vr + 8);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const float* Vr = (const float*)Vb + (long)(32 * d + l31) * a.vwt_ld + key0 + 4 * half;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const float4 v = *(const float4*)(Vr + 8 * gq);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, s[4 * gq + 0], o[d], 0, 0, 0);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, s[4 * gq + 1], o[d], 0, 0, 0);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, s[4 * gq + 2], o[d], 0, 0, 0);
          o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, s[4 * gq + 3], o[d], 0, 0, 0);
        }
      }
    }
  }
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (q >= a.N) return;
  TOUT* Y = a.out ? (TOUT*)a.out + (long)b * a.out_bs + (long)q * a.out_ld + h * 64 : nullptr;
  TOUT* Z = a.out_act ? (TOUT*)a.out_act + (long)b * a.act_bs + (long)q * a.act_ld + h * 64 : nullptr;
  const TOUT* R = a.resid ? (const TOUT*)a.resid + (long)b * a.resid_bs + (long)q * a.resid_ld + h * 64 : nullptr;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int dv = 32 * d + 8 * gq + 4 * half;
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        y[e] = o[d][4 * gq + e] * inv;
        if (a.bout) y[e] += a.bout[h * 64 + dv + e];
      }
      if constexpr (sizeof(TOUT) == 4) {
        if (Y) *(float4*)((float*)Y + dv) = make_float4(y[0], y[1], y[2], y[3]);
        if (Z) {
          float4 rv = R ? *(const float4*)((const float*)R + dv) : make_float4(0, 0, 0, 0);
          *(float4*)((float*)Z + dv) = make_float4(fmaxf(rv.x + y[0], 0.f), fmaxf(rv.y + y[1], 0.f),
                                                    fmaxf(rv.z + y[2], 0.f), fmaxf(rv.w + y[3], 0.f));
        }
      } else {
        if (Y) *(uint2*)((unsigned short*)Y + dv) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
        if (Z) {
          float rr[4] = {0, 0, 0, 0};
          if (R) {
            const uint2 rv = *(const uint2*)((const unsigned short*)R + dv);
            rr[0] = bf2f(rv.x & 0xffff); rr[1] = bf2f(rv.x >> 16);
            rr[2] = bf2f(rv.y & 0xffff); rr[3] = bf2f(rv.y >> 16);
          }
          *(uint2*)((unsigned short*)Z + dv) =
              make_uint2(pack_bf16x2(fmaxf(rr[0] + y[0], 0.f), fmaxf(rr[1] + y[1], 0.f)),
                         pack_bf16x2(fmaxf(rr[2] + y[2], 0.f), fmaxf(rr[3] + y[3], 0.f)));
        }
      }
    }
}


// ---------------------------------------------------------------------------------------
// bf16 throughput kernel: one WORKGROUP = all query tiles of one (image, head) (up to 16
// wavefronts, 32 queries each).  The head's K rows and VW^T columns are staged ONCE per key
// chunk of 320 keys in LDS and shared by every wavefront (the v1 kernel above re-reads them
// from L2 per wavefront and is latency bound: 3.7 % MFMA utilisation in profiles/r01).  The
// geometry bias arrives as fp16; the epilogue transposes O^T through LDS so that the output,
// residual and activation rows move as 16-byte coalesced accesses.
// ---------------------------------------------------------------------------------------
constexpr int kKC = 320;                    // keys per LDS chunk
constexpr int kOLD = 72;                    // epilogue row stride (bf16): 64 + 8 pad

// lds layout: sK [kc_rows][64] bf16 (16-B chunks XOR-swizzled), sV [64][vld] bf16 (+ slack); the
// epilogue reuses the start of the buffer as [nwave][32][kOLD] bf16.
// BIAS_F32 (round 6): the bias is float32 ln G (the training forward's geometry, which its backward re-derives the softmax from) instead of fp16 log2 G:
// 16-byte loads per four keys, multiplied by log2 e on the way into the log2-domain logits.
template <bool BIAS_F32>
__global__ __launch_bounds__(1024) void relation_attention_lds_kernel(AttnArgs a, int kc_rows, int vld) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* sK = smem;
  unsigned short* sV = (unsigned short*)(smem + kc_rows * 128);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const int qt = blockIdx.x * nwave + wave;
  const bool wave_on = qt * 32 < a.N;
  const int q = qt * 32 + l31;
  const int qc = q < a.N ? q : a.N - 1;
  typedef unsigned short T;
  const T* Q = (const T*)a.q + (long)b * a.q_bs + (long)qc * a.q_ld + h * 64;
  const T* Kb = (const T*)a.k + (long)b * a.k_bs + h * 64;
  const T* Vb = (const T*)a.vwt + (long)b * a.vwt_bs + (long)(h * 64) * a.vwt_ld;
  typedef typename std::conditional<BIAS_F32, float, __half>::type TBIAS;
  typedef typename std::conditional<BIAS_F32, float4, uint2>::type BV;      // four keys' bias values
  const TBIAS* Bq = (const TBIAS*)a.bias + (long)b * a.bias_bs + ((long)h * a.N + qc) * a.Mpad;

  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(Q + 16 * kk + 8 * half);
  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float scale2 = a.scale * 1.44269504088896340736f;
  const int Mb = a.key_count ? min(max(a.key_count[b], 1), a.M) : a.M;      // keys of THIS image

  for (int kc0 = 0; kc0 < a.M; kc0 += kKC) {
    if (kc0 > 0) __syncthreads();
    const int clen = min(a.M - kc0, kKC);             // valid keys of this chunk
    // bias of the first tile: issued before the staging so that its latency overlaps it
    BV bcur[4];
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) bcur[gq] = *(const BV*)(Bq + kc0 + 8 * gq + 4 * half);
    // ---- stage K rows [kc0, kc0+clen) and VW^T columns of this head --------------------------
    // (batches of 2 / 4 loads issued before their LDS writes -- more would spill at the 128 registers of a 1024-thread launch bound: a load -> write loop costs one L2 round trip per iteration, and with 124 VGPRs only ONE
    //  10-wave workgroup is resident per CU, so nothing else runs meanwhile -- r05: the staging was ~2/3 of a workgroup's 30 us)
    for (int c0 = tid; c0 < clen * 8; c0 += 2 * nthr) {
      uint4 kq[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = min(c0 + i * nthr, clen * 8 - 1), row = c >> 3, ch = c & 7;          // (clamped, unconditional: a guarded refill keeps kq in scratch)
        kq[i] = *(const uint4*)(Kb + (long)(kc0 + row) * a.k_ld + ch * 8);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int c = c0 + i * nthr, row = c >> 3, ch = c & 7;
        if (c < clen * 8) *(uint4*)(sK + row * 128 + ((ch ^ (row & 7)) << 4)) = kq[i];
      }
    }
    const int v4 = (clen + 3) >> 2;                   // 4-key groups per row
    // zero what the masked tail of the last tile may touch (P is 0 there, but 0 x NaN is not)
    if (kc0 == 0) {
      for (int c = tid; c < 64 * ((vld >> 2) - v4); c += nthr) {
        const int row = c / ((vld >> 2) - v4), c4 = v4 + c % ((vld >> 2) - v4);
        *(uint2*)(sV + row * vld + 4 * c4) = make_uint2(0, 0);
      }
      if (tid < 16) *(uint2*)(sV + 64 * vld + 4 * tid) = make_uint2(0, 0);
    }
    for (int c0 = tid; c0 < 64 * v4; c0 += 4 * nthr) {
      uint2 vq[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = min(c0 + i * nthr, 64 * v4 - 1), row = c / v4, c4 = c - row * v4;
        vq[i] = *(const uint2*)(Vb + (long)row * a.vwt_ld + kc0 + 4 * c4);   // pad cols are 0
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int c = c0 + i * nthr, row = c / v4, c4 = c - row * v4;
        if (c < 64 * v4) *(uint2*)(sV + row * vld + 4 * c4) = vq[i];
      }
    }
    __syncthreads();
    if (wave_on) {
      const int ntile = (clen + 31) / 32;
      for (int kt = 0; kt < ntile; ++kt) {
        const int key0 = kc0 + kt * 32;
        // prefetch the next tile's bias (clamped: the last prefetch re-reads a valid address)
        BV bnext[4];
        {
          const int kn = (kt + 1 < ntile) ? key0 + 32 : key0;
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) bnext[gq] = *(const BV*)(Bq + kn + 8 * gq + 4 * half);
        }
        f32x16 s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.f;
        {
          int row = kt * 32 + l31;
          row = row < clen ? row : clen - 1;          // keys past M: any valid row, masked below
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const bf16x8 kf = *(const bf16x8*)(sK + row * 128 + (((2 * kk + half) ^ (row & 7)) << 4));
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kk], s, 0, 0, 0);
          }
        }
        // logits in the log2 domain: v = log2(G) + (scale * log2 e) * (q . k)
        float tmax = -INFINITY;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          float bb[4];
          if constexpr (BIAS_F32) {
            const float4 bv = *(const float4*)&bcur[gq];
            bb[0] = bv.x * 1.44269504088896340736f; bb[1] = bv.y * 1.44269504088896340736f; bb[2] = bv.z * 1.44269504088896340736f; bb[3] = bv.w * 1.44269504088896340736f;
          } else {
            const uint2 bu = *(const uint2*)&bcur[gq];
            const __half2 b01 = *(const __half2*)&bu.x, b23 = *(const __half2*)&bu.y;
            bb[0] = __low2float(b01); bb[1] = __high2float(b01); bb[2] = __low2float(b23); bb[3] = __high2float(b23);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int r = 4 * gq + e;
            s[r] = fmaf(s[r], scale2, bb[e]);
          }
        }
        if (key0 + 32 > Mb) {                            // only the tiles that hold keys past the image's count
#pragma unroll
          for (int gq = 0; gq < 4; ++gq)
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (key0 + 8 * gq + 4 * half + e >= Mb) s[4 * gq + e] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        // deferred rescale: the running maximum only moves when some query's tile maximum exceeds it
        // by more than 2^8; P is then bounded by 256 instead of 1 (exact after the final 1/l)
        if (__any(tmax > m_run + 8.0f)) {
          const float m_new = fmaxf(m_run, tmax);
          const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);     // first tile: exp2(-inf) = 0
          l_run *= alpha;
          m_run = m_new;
#pragma unroll
          for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[r] - m_run);
          s[r] = pv;
          psum += pv;
        }
        l_run += psum;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          bf16x8 pf;
          unsigned int* pw = (unsigned int*)&pf;
#pragma unroll
          for (int t = 0; t < 4; ++t) pw[t] = pack_bf16x2(s[8 * ks + 2 * t], s[8 * ks + 2 * t + 1]);
#pragma unroll
          for (int d = 0; d < 2; ++d) {
            // columns past the chunk hold finite data of the next row (or the slack): P is 0 there
            const unsigned short* vr = sV + (32 * d + l31) * vld + kt * 32 + 16 * ks + 4 * half;
            bf16x8 vf;
            *(uint2*)&vf = *(const uint2*)vr;
            *((uint2*)&vf + 1) = *(const uint2*)(vr + 8);
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[d], 0, 0, 0);
          }
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) bcur[gq] = bnext[gq];
      }
    }
  }
  // ---- epilogue: (O^T / l + bout) -> bf16 -> LDS [q][dv] -> 16-byte coalesced rows -------------
  __syncthreads();
  unsigned short* so = (unsigned short*)smem + wave * (32 * kOLD);
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  if (wave_on) {
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int dv = 32 * d + 8 * gq + 4 * half;
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = o[d][4 * gq + e] * inv + (a.bout ? a.bout[h * 64 + dv + e] : 0.f);
        *(uint2*)(so + l31 * kOLD + dv) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
      }
    // same wave reads back what it wrote: no workgroup barrier needed, only LDS completion
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    unsigned short* Y = a.out ? (unsigned short*)a.out + (long)b * a.out_bs + h * 64 : nullptr;
    unsigned short* Z = a.out_act ? (unsigned short*)a.out_act + (long)b * a.act_bs + h * 64 : nullptr;
    const unsigned short* R = a.resid ? (const unsigned short*)a.resid + (long)b * a.resid_bs + h * 64 : nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = lane + 64 * i, qq = idx >> 3, c8 = (idx & 7) * 8;
      const int qrow = qt * 32 + qq;
      if (qrow >= a.N) continue;
      const uint4 yv = *(const uint4*)(so + qq * kOLD + c8);
      if (Y) *(uint4*)(Y + (long)qrow * a.out_ld + c8) = yv;
      if (Z) {
        // relu(resid + Y) on the bf16 attention output, i.e. exactly the unfused bf16 graph
        const unsigned int yw[4] = {yv.x, yv.y, yv.z, yv.w};
        unsigned int rw[4] = {0u, 0u, 0u, 0u};
        if (R) {
          const uint4 rv = *(const uint4*)(R + (long)qrow * a.resid_ld + c8);
          rw[0] = rv.x; rw[1] = rv.y; rw[2] = rv.z; rw[3] = rv.w;
        }
        unsigned int zw[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
          zw[e] = pack_bf16x2(fmaxf(bf2f(yw[e] & 0xffff) + bf2f(rw[e] & 0xffff), 0.f),
                              fmaxf(bf2f(yw[e] >> 16) + bf2f(rw[e] >> 16), 0.f));
        *(uint4*)(Z + (long)qrow * a.act_ld + c8) = make_uint4(zw[0], zw[1], zw[2], zw[3]);
      }
    }
  }
}



// ---------------------------------------------------------------------------------------
// Fused relation module (bf16 throughput path): geometry bias + attention in ONE kernel; the
// [B,16,N,Mpad] bias tensor never exists in HBM (322 MB written + 2 x 166 MB read per step at
// B = 54 with the two-kernel path).
//
// One workgroup = 32 queries of one image x ALL 16 heads (wave w = head w), so the 64 sin/cos
// features of a (query, key) pair are evaluated once for the 16 heads that consume them:
//   phase G (key tile t+1): wave w takes queries 2w, 2w+1 of the tile against the 32 keys, as four
//       16-pair sets.  Lane (col = lane & 15, grp = lane >> 4) owns pair `col` and the 8 wavelengths
//       of ONE (coordinate, sin|cos) combination per k-step: axis = grp >> 1 (x|y), sc = grp & 1, k-step
//       0 = centre distance, 1 = log size ratio -- exactly the B-operand layout of
//       mfma_f32_16x16x32_f16 for feature f = 32 kk + 8 grp + t (SYM_REL:29-44: f = 16 c + 8 sc + t).
//       pair_pos_fc1 (64 -> 16, SYM_REL:109-116) is two MFMAs against the fp16 weight fragments
//       (A operand, rows = heads); log2(max(relu(.), 1e-6)) of the 16 x 16 result goes to LDS as
//       fp32 [head][query][key] (double buffered: one barrier per key tile).
//   phase A (key tile t): wave h runs the flash-style update of relation_attention_lds_kernel for its
//       head, bias from LDS, K rows / VW^T columns straight from L2 (no other wave shares them; the
//       workgroup -> image mapping keeps the 10 query tiles of an image on one XCD so that the
//       re-reads hit that XCD's L2).
// Precision: fp16 features and weights in the 64 -> 16 product (|feature| <= 1: 2^-11 absolute;
// measured against the fp32 geometry kernel in tests/test_gpu_relation.py) -- tighter than the fp16
// rounding of log2 G the two-kernel path stores.
// ---------------------------------------------------------------------------------------
constexpr int kGQ = 36;                    // floats per query row of the LDS bias tile (32 keys + 4: conflict-free float4 reads)
constexpr int kGH = 32 * kGQ + 4;          // floats per head (+4: the four lane groups of a store hit distinct banks)
constexpr int kGBuf = 16 * kGH;            // floats per buffer

struct FusedArgs {
  AttnArgs a;
  const float* boxes; int box_stride, box_off;
  const float* wp;                         // [16][64] fp32 pair_pos_fc1 weight of this module
  const float* bp;                         // [16]
  float c2[8];                             // ln2 * 100 / (2 pi * wave_length^(t/8)): log2-domain position -> revolutions
  int nq;                                  // query tiles per image
};

__global__ __launch_bounds__(1024) void relation_fused_kernel(FusedArgs f) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const AttnArgs& a = f.a;
  float* sG = (float*)smem;                               // [2][16][kGH]
  float* sBox = sG + 2 * kGBuf;                           // [ntile * 32][4]: cx, cy, log2 w, log2 h of the keys
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  const int col = lane & 15, grp = lane >> 4, axis = grp >> 1;
  // images in groups of 8, one per XCD (consecutive workgroup ids go round-robin over the 8 XCDs)
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int b = (slot / f.nq) * 8 + xcd, qt = slot % f.nq;
  if (b >= a.B) return;
  const int h = wave;
  const int ntile = (a.M + 31) / 32;
  typedef unsigned short T;
  const float* bx = f.boxes + (long)b * a.N * f.box_stride + f.box_off;

  // ---- key table ------------------------------------------------------------------------------
  for (int j = tid; j < ntile * 32; j += 1024) {
    const float* pj = bx + (long)(j < a.M ? j : a.M - 1) * f.box_stride;
    const float w = pj[2] - pj[0] + 1.f, hh = pj[3] - pj[1] + 1.f;
    *(float4*)(sBox + 4 * j) = make_float4(0.5f * (pj[0] + pj[2]), 0.5f * (pj[1] + pj[3]),
                                            __builtin_amdgcn_logf(w), __builtin_amdgcn_logf(hh));
  }
  // ---- this wave's two geometry queries (phase G) ---------------------------------------------------
  float cq[2], isq[2], l2q[2];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    int qi = qt * 32 + 2 * wave + qs;
    qi = qi < a.N ? qi : a.N - 1;
    const float* pi = bx + (long)qi * f.box_stride;
    const float lo = pi[axis], hi = pi[2 + axis];
    const float sz = hi - lo + 1.f;
    cq[qs] = 0.5f * (lo + hi); isq[qs] = 1.0f / sz; l2q[qs] = __builtin_amdgcn_logf(sz);
  }
  // pair_pos_fc1 fragments: A operand, row = head (lane & 15), k = 32 kk + 8 grp + t
  f16x8 wf[2];
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const float* wr = f.wp + col * 64 + 32 * kk + 8 * grp;
    const float4 w0 = *(const float4*)wr, w1 = *(const float4*)(wr + 4);
    wf[kk][0] = (_Float16)w0.x; wf[kk][1] = (_Float16)w0.y; wf[kk][2] = (_Float16)w0.z; wf[kk][3] = (_Float16)w0.w;
    wf[kk][4] = (_Float16)w1.x; wf[kk][5] = (_Float16)w1.y; wf[kk][6] = (_Float16)w1.z; wf[kk][7] = (_Float16)w1.w;
  }
  f32x4 bp4;
#pragma unroll
  for (int e = 0; e < 4; ++e) bp4[e] = f.bp[4 * grp + e];
  const float phase = (grp & 1) ? 0.25f : 0.f;            // cos x = sin(x + 1/4 revolution)

  auto geometry_tile = [&](int kt, float* gb) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int key = kt * 32 + 16 * ks + col;
      const float ck = sBox[4 * key + axis], l2k = sBox[4 * key + 2 + axis];
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
        const float d = fmaxf(fabsf(cq[qs] - ck) * isq[qs], 1e-3f);       // SYM_REL:59-66
        const float Ld = __builtin_amdgcn_logf(d);                          // log2; ln 2 is folded into c2
        const float Ls = l2q[qs] - l2k;                                     // log2(w_i / w_j), SYM_REL:67-70
        u32x4 w0, w1;                                                       // 8 fp16 features each
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          w0[t >> 1] = pack_f16x2_rn(__builtin_amdgcn_sinf(fmaf(Ld, f.c2[t], phase)), __builtin_amdgcn_sinf(fmaf(Ld, f.c2[t + 1], phase)));
          w1[t >> 1] = pack_f16x2_rn(__builtin_amdgcn_sinf(fmaf(Ls, f.c2[t], phase)), __builtin_amdgcn_sinf(fmaf(Ls, f.c2[t + 1], phase)));
        }
        const f16x8 e0 = __builtin_bit_cast(f16x8, w0), e1 = __builtin_bit_cast(f16x8, w1);
        f32x4 acc = bp4;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[0], e0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[1], e1, acc, 0, 0, 0);
        float* gp = gb + (4 * grp) * kGH + (2 * wave + qs) * kGQ + 16 * ks + col;
#pragma unroll
        for (int e = 0; e < 4; ++e) gp[e * kGH] = __builtin_amdgcn_logf(fmaxf(acc[e], 1e-6f));   // relu, clamp, log (SYM_REL:116,139)
      }
    }
  };

  // ---- attention state of (head h, queries qt*32 + l31) ---------------------------------------------
  const int q = qt * 32 + l31;
  const int qc = q < a.N ? q : a.N - 1;
  const T* Q = (const T*)a.q + (long)b * a.q_bs + (long)qc * a.q_ld + h * 64;
  const T* Kb = (const T*)a.k + (long)b * a.k_bs + h * 64;
  const T* Vb = (const T*)a.vwt + (long)b * a.vwt_bs + (long)(h * 64) * a.vwt_ld;
  bf16x8 qf[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) qf[kk] = *(const bf16x8*)(Q + 16 * kk + 8 * half);
  f32x16 o[2];
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float scale2 = a.scale * 1.44269504088896340736f;

  __syncthreads();                                         // key table
  geometry_tile(0, sG);
  __syncthreads();
  for (int kt = 0; kt < ntile; ++kt) {
    const int key0 = kt * 32;
    const float* gb = sG + (kt & 1) * kGBuf + h * kGH + l31 * kGQ + 4 * half;
    // K rows of this tile: requested before the geometry of the next tile so that their latency is covered
    bf16x8 kf[4];
    {
      int kr = key0 + l31;
      kr = kr < a.M ? kr : a.M - 1;                       // keys past M: any valid row, masked below
      const T* Kr = Kb + (long)kr * a.k_ld + 8 * half;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[kk] = *(const bf16x8*)(Kr + 16 * kk);
    }
    if (kt + 1 < ntile) geometry_tile(kt + 1, sG + ((kt + 1) & 1) * kGBuf);
    bf16x8 vf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int d = 0; d < 2; ++d) {
        const T* Vr = Vb + (long)(32 * d + l31) * a.vwt_ld + key0 + 16 * ks + 4 * half;   // pad columns are 0
        *(uint2*)&vf[ks][d] = *(const uint2*)Vr;
        *((uint2*)&vf[ks][d] + 1) = *(const uint2*)(Vr + 8);
      }
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], s, 0, 0, 0);
    // logits in the log2 domain: v = log2(G) + (scale * log2 e) * (q . k)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const float4 bv = *(const float4*)(gb + 8 * gq);
      s[4 * gq + 0] = fmaf(s[4 * gq + 0], scale2, bv.x);
      s[4 * gq + 1] = fmaf(s[4 * gq + 1], scale2, bv.y);
      s[4 * gq + 2] = fmaf(s[4 * gq + 2], scale2, bv.z);
      s[4 * gq + 3] = fmaf(s[4 * gq + 3], scale2, bv.w);
    }
    if (key0 + 32 > a.M) {                                // only the last tile has keys past M
#pragma unroll
      for (int gq = 0; gq < 4; ++gq)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (key0 + 8 * gq + 4 * half + e >= a.M) s[4 * gq + e] = -INFINITY;
    }
    float tmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) tmax = fmaxf(tmax, s[r]);
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
    if (__any(tmax > m_run + 8.0f)) {                     // deferred rescale, as in relation_attention_lds_kernel
      const float m_new = fmaxf(m_run, tmax);
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      l_run *= alpha;
      m_run = m_new;
#pragma unroll
      for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pv = __builtin_amdgcn_exp2f(s[r] - m_run);
      s[r] = pv;
      psum += pv;
    }
    l_run += psum;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 pf;
      unsigned int* pw = (unsigned int*)&pf;
#pragma unroll
      for (int t = 0; t < 4; ++t) pw[t] = pack_bf16x2(s[8 * ks + 2 * t], s[8 * ks + 2 * t + 1]);
#pragma unroll
      for (int d = 0; d < 2; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[ks][d], pf, o[d], 0, 0, 0);
    }
    __syncthreads();                                      // tile kt consumed by every head, tile kt+1 complete
  }
  // ---- epilogue: (O^T / l + bout) -> bf16 -> LDS [q][dv] -> 16-byte coalesced rows (the bias buffers are free now)
  unsigned short* so = (unsigned short*)smem + wave * (32 * kOLD);
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
#pragma unroll
  for (int d = 0; d < 2; ++d)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int dv = 32 * d + 8 * gq + 4 * half;
      float y[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = o[d][4 * gq + e] * inv + (a.bout ? a.bout[h * 64 + dv + e] : 0.f);
      *(uint2*)(so + l31 * kOLD + dv) = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
    }
  __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0): the same wave reads back what it wrote
  __builtin_amdgcn_wave_barrier();
  unsigned short* Y = a.out ? (unsigned short*)a.out + (long)b * a.out_bs + h * 64 : nullptr;
  unsigned short* Z = a.out_act ? (unsigned short*)a.out_act + (long)b * a.act_bs + h * 64 : nullptr;
  const unsigned short* R = a.resid ? (const unsigned short*)a.resid + (long)b * a.resid_bs + h * 64 : nullptr;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = lane + 64 * i, qq = idx >> 3, c8 = (idx & 7) * 8;
    const int qrow = qt * 32 + qq;
    if (qrow >= a.N) continue;
    const uint4 yv = *(const uint4*)(so + qq * kOLD + c8);
    if (Y) *(uint4*)(Y + (long)qrow * a.out_ld + c8) = yv;
    if (Z) {
      const unsigned int yw[4] = {yv.x, yv.y, yv.z, yv.w};
      unsigned int rw[4] = {0u, 0u, 0u, 0u};
      if (R) {
        const uint4 rv = *(const uint4*)(R + (long)qrow * a.resid_ld + c8);
        rw[0] = rv.x; rw[1] = rv.y; rw[2] = rv.z; rw[3] = rv.w;
      }
      unsigned int zw[4];
#pragma unroll
      for (int e = 0; e < 4; ++e)
        zw[e] = pack_bf16x2(fmaxf(bf2f(yw[e] & 0xffff) + bf2f(rw[e] & 0xffff), 0.f),
                            fmaxf(bf2f(yw[e] >> 16) + bf2f(rw[e] >> 16), 0.f));
      *(uint4*)(Z + (long)qrow * a.act_ld + c8) = make_uint4(zw[0], zw[1], zw[2], zw[3]);
    }
  }
}

}  // namespace relnet

using namespace relnet;
enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

extern "C" int relnet_geometry_bias(const float* boxes, int box_stride, int box_off,
                                    const float* wp, const float* bp, const float* divisors8,
                                    void* bias, int bias_half, float* pos_mat, float* pos_emb, int B, int N,
                                    int M, int Mpad, int fc_dim, int nmod, void* stream) {
  RELNET_REQUIRE(boxes && wp && bp && divisors8 && bias, "relnet_geometry_bias: null operand");
  RELNET_REQUIRE(fc_dim == 16, "relnet_geometry_bias: fc_dim %d unsupported (16 only)", fc_dim);
  RELNET_REQUIRE(nmod == 1 || nmod == 2, "relnet_geometry_bias: nmod %d unsupported (1 or 2)", nmod);
  RELNET_REQUIRE(B > 0 && N > 0 && M > 0 && M <= N && Mpad >= M && Mpad % 32 == 0,
                 "relnet_geometry_bias: bad shape B=%d N=%d M=%d Mpad=%d", B, N, M, Mpad);
  GeomArgs g;
  g.boxes = boxes; g.box_stride = box_stride; g.box_off = box_off; g.wp = wp; g.bp = bp;
  for (int k = 0; k < 8; ++k) g.divisors[k] = divisors8[k];
  g.bias = bias; g.pos_mat = pos_mat; g.pos_emb = pos_emb;
  g.B = B; g.N = N; g.M = M; g.Mpad = Mpad; g.nmod = nmod;
  dim3 grid((unsigned)(((long)N * M + 255) / 256), B);
  hipStream_t s = (hipStream_t)stream;
  for (int k = 0; k < 8; ++k) g.c2[k] = (float)(0.69314718055994530942 * 100.0 / (6.283185307179586476925 * (double)divisors8[k]));
  const size_t key_lds = (size_t)((M + 63) / 64) * 64 * 16;
  RELNET_REQUIRE(bias_half >= -1 && bias_half <= 2, "relnet_geometry_bias: bias_half %d (0: float32 exact, 1: fp16 log2, 2: float32 ln from the matrix-core kernel, -1: float32 libm)", bias_half);
  g.out_f32 = bias_half == 2 ? 1 : 0;
  if (bias_half == 2 && (pos_mat || pos_emb || key_lds > 64 * 1024)) bias_half = -1;      // shapes the matrix-core kernel does not take
  if ((bias_half == 1 || bias_half == 2) && !pos_mat && !pos_emb && key_lds <= 64 * 1024) {
    // throughput path: pair_pos_fc1 of all modules on the matrix cores, one wavefront per query
    dim3 g2((unsigned)((N + 3) / 4), B);
    geometry_bias_mfma_kernel<<<g2, 256, key_lds, s>>>(g);
  } else if (bias_half == 1) {
    if (nmod == 1) geometry_bias_kernel<16, 1, __half><<<grid, 256, 0, s>>>(g);
    else geometry_bias_kernel<16, 2, __half><<<grid, 256, 0, s>>>(g);
  } else if (bias_half == -1) {    // float32 output, float32 libm arithmetic: the training backward's recompute of log G
    if (nmod == 1) geometry_bias_kernel<16, 1, float><<<grid, 256, 0, s>>>(g);
    else geometry_bias_kernel<16, 2, float><<<grid, 256, 0, s>>>(g);
  } else if (bias_half == 0) {      // float32 parity path: oracle arithmetic (float64 sin / cos / log, float64 accumulation)
    if (nmod == 1) geometry_bias_kernel<16, 1, float, true><<<grid, 256, 0, s>>>(g);
    else geometry_bias_kernel<16, 2, float, true><<<grid, 256, 0, s>>>(g);
  }
  return check_launch("relnet_geometry_bias");
}

static int g_attn_lds_f32 = 1;     // tuning / test knob: 1 = bf16 attention with a float32 bias runs on the LDS kernel, 0 = on the streaming kernel (rounds 1 - 5)
extern "C" void relnet_relation_attention_debug_lds_f32(int on) { g_attn_lds_f32 = on; }

extern "C" int relnet_relation_attention_kc(
    const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs, const void* vwt,
    long vwt_ld, long vwt_bs, const void* bias, int bias_half, long bias_bs, const float* bout,
    const void* resid, long resid_ld, long resid_bs, void* out, long out_ld, long out_bs,
    void* out_act, long act_ld, long act_bs, float* logits, int B, int H, int N, int M, int Mpad,
    float scale, int in_dtype, int out_dtype, const int* key_count, void* stream) {
  RELNET_REQUIRE(q && k && vwt && bias, "relnet_relation_attention: null operand");
  RELNET_REQUIRE(out || out_act, "relnet_relation_attention: no output requested");
  RELNET_REQUIRE(B > 0 && H > 0 && N > 0 && M > 0 && Mpad >= M && Mpad % 32 == 0,
                 "relnet_relation_attention: bad shape B=%d H=%d N=%d M=%d Mpad=%d", B, H, N, M, Mpad);
  RELNET_REQUIRE(in_dtype == out_dtype, "relnet_relation_attention: in/out dtype must match (%d vs %d)", in_dtype, out_dtype);
  AttnArgs a;
  a.q = q; a.q_ld = q_ld; a.q_bs = q_bs; a.k = k; a.k_ld = k_ld; a.k_bs = k_bs;
  a.vwt = vwt; a.vwt_ld = vwt_ld; a.vwt_bs = vwt_bs; a.bias = bias; a.bias_bs = bias_bs;
  a.bout = bout; a.resid = resid; a.resid_ld = resid_ld; a.resid_bs = resid_bs;
  a.out = out; a.out_ld = out_ld; a.out_bs = out_bs; a.out_act = out_act; a.act_ld = act_ld;
  a.act_bs = act_bs; a.logits = logits; a.B = B; a.H = H; a.N = N; a.M = M; a.Mpad = Mpad;
  a.scale = scale; a.key_count = key_count;
  dim3 grid((unsigned)(((N + 31) / 32 + 3) / 4), H, B);
  hipStream_t s = (hipStream_t)stream;
  // round 6: the LDS kernel also takes the float32 ln G of the training forward (g_attn_lds_f32 = 0: the streaming kernel as before)
  if (in_dtype == RELNET_BF16 && (bias_half || (g_attn_lds_f32 && !logits && Mpad % 4 == 0 && q_ld % 8 == 0 && k_ld % 8 == 0 && vwt_ld % 4 == 0 && out_ld % 8 == 0 &&
                                                   act_ld % 8 == 0 && resid_ld % 8 == 0 && (((uintptr_t)bias) & 15) == 0 && bias_bs % 4 == 0))) {
    // LDS kernel: fp16 log2 G (or float32 ln G) bias, no logits output; one workgroup per (image, head, <=16 query tiles)
    RELNET_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vwt_ld % 4 == 0 && Mpad % 4 == 0, "relnet_relation_attention(bf16): row strides must be 16-byte (q,k) / 8-byte (vwt) aligned");
    RELNET_REQUIRE(!logits, "relnet_relation_attention: logits output needs the fp32-bias kernel (bias_half = 0)");
    RELNET_REQUIRE(out_ld % 8 == 0 && act_ld % 8 == 0 && resid_ld % 8 == 0, "relnet_relation_attention(bf16): output rows must be 16-byte aligned");
    const int qtiles = (N + 31) / 32;
    // one workgroup = all query tiles of the (image, head) (measured: splitting it in two to overlap
    // staging with compute loses more to the duplicated K / VW^T staging than it gains)
    int nwave = qtiles < 16 ? qtiles : 16;
    if (const char* e = getenv("RELNET_ATTN_WAVES")) nwave = atoi(e) > 0 && atoi(e) <= 16 ? atoi(e) : nwave;
    const int kc_rows = M < kKC ? M : kKC;                         // K rows staged per chunk
    int vk = (kc_rows + 3) / 4 + 1;                                 // VW^T row stride (bf16) = 4 * odd:
    if ((vk & 1) == 0) ++vk;                                        // 32 rows x 8-byte reads hit 32 distinct bank pairs
    const int vld = 4 * vk;
    const size_t kv = (size_t)kc_rows * 128 + (size_t)64 * vld * 2 + 128 /* slack for the masked tail */;
    const size_t ep = (size_t)nwave * 32 * kOLD * 2;
    const size_t lds = kv > ep ? kv : ep;
    static relnet::PerDeviceOnce attr_once;
    if (attr_once.first()) {
      hipFuncSetAttribute((const void*)relation_attention_lds_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipFuncSetAttribute((const void*)relation_attention_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    dim3 g2((unsigned)((qtiles + nwave - 1) / nwave), H, B);
    if (bias_half) relation_attention_lds_kernel<false><<<g2, nwave * 64, lds, s>>>(a, kc_rows, vld);
    else relation_attention_lds_kernel<true><<<g2, nwave * 64, lds, s>>>(a, kc_rows, vld);
  } else if (in_dtype == RELNET_BF16) {
    RELNET_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vwt_ld % 4 == 0, "relnet_relation_attention(bf16): row strides must be 16-byte (q,k) / 8-byte (vwt) aligned");
    relation_attention_kernel<unsigned short, unsigned short><<<grid, 256, 0, s>>>(a);
  } else if (in_dtype == RELNET_F32) {
    RELNET_REQUIRE(!bias_half, "relnet_relation_attention(f32): fp32 bias required");
    RELNET_REQUIRE(q_ld % 4 == 0 && k_ld % 4 == 0 && vwt_ld % 4 == 0, "relnet_relation_attention(f32): row strides must be 16-byte aligned");
    relation_attention_kernel<float, float><<<grid, 256, 0, s>>>(a);
  } else {
    RELNET_REQUIRE(false, "relnet_relation_attention: unknown dtype %d", in_dtype);
  }
  return check_launch("relnet_relation_attention");
}

extern "C" int relnet_relation_attention(
    const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs, const void* vwt,
    long vwt_ld, long vwt_bs, const void* bias, int bias_half, long bias_bs, const float* bout,
    const void* resid, long resid_ld, long resid_bs, void* out, long out_ld, long out_bs,
    void* out_act, long act_ld, long act_bs, float* logits, int B, int H, int N, int M, int Mpad,
    float scale, int in_dtype, int out_dtype, void* stream) {
  return relnet_relation_attention_kc(q, q_ld, q_bs, k, k_ld, k_bs, vwt, vwt_ld, vwt_bs, bias, bias_half, bias_bs, bout, resid,
                                      resid_ld, resid_bs, out, out_ld, out_bs, out_act, act_ld, act_bs, logits, B, H, N, M, Mpad,
                                      scale, in_dtype, out_dtype, nullptr, stream);
}

// Fused geometry + attention of one relation module (bf16; H = 16 heads x 64; M <= 640 keys).
// boxes [B][N][box_stride] fp32 (xyxy at +box_off), wp [16][64] / bp [16] = pair_pos_fc1 of this module,
// divisors8 = wave_length^(t/8) (host pointer).  Replaces relnet_geometry_bias + relnet_relation_attention
// (SYM_REL:29-83 + :109-150) on the throughput path; outputs as relnet_relation_attention.
extern "C" int relnet_relation_attention_fused(
    const void* q, long q_ld, long q_bs, const void* k, long k_ld, long k_bs, const void* vwt,
    long vwt_ld, long vwt_bs, const float* boxes, int box_stride, int box_off, const float* wp,
    const float* bp, const float* divisors8, const float* bout, const void* resid, long resid_ld,
    long resid_bs, void* out, long out_ld, long out_bs, void* out_act, long act_ld, long act_bs,
    int B, int H, int N, int M, int Mpad, float scale, void* stream) {
  RELNET_REQUIRE(q && k && vwt && boxes && wp && bp && divisors8, "relnet_relation_attention_fused: null operand");
  RELNET_REQUIRE(out || out_act, "relnet_relation_attention_fused: no output requested");
  RELNET_REQUIRE(H == 16, "relnet_relation_attention_fused: %d heads unsupported (16 only: one wavefront per head)", H);
  RELNET_REQUIRE(B > 0 && N > 0 && M > 0 && M <= N && M <= 640 && Mpad >= M && Mpad % 32 == 0,
                 "relnet_relation_attention_fused: bad shape B=%d N=%d M=%d Mpad=%d (M <= 640)", B, N, M, Mpad);
  RELNET_REQUIRE(q_ld % 8 == 0 && k_ld % 8 == 0 && vwt_ld % 4 == 0 && out_ld % 8 == 0 && act_ld % 8 == 0 && resid_ld % 8 == 0,
                 "relnet_relation_attention_fused: row strides must be 16-byte (q, k, outputs) / 8-byte (vwt) aligned");
  FusedArgs f;
  AttnArgs& a = f.a;
  a.q = q; a.q_ld = q_ld; a.q_bs = q_bs; a.k = k; a.k_ld = k_ld; a.k_bs = k_bs;
  a.vwt = vwt; a.vwt_ld = vwt_ld; a.vwt_bs = vwt_bs; a.bias = nullptr; a.bias_bs = 0;
  a.bout = bout; a.resid = resid; a.resid_ld = resid_ld; a.resid_bs = resid_bs;
  a.out = out; a.out_ld = out_ld; a.out_bs = out_bs; a.out_act = out_act; a.act_ld = act_ld;
  a.act_bs = act_bs; a.logits = nullptr; a.B = B; a.H = H; a.N = N; a.M = M; a.Mpad = Mpad; a.scale = scale; a.key_count = nullptr;
  f.boxes = boxes; f.box_stride = box_stride; f.box_off = box_off; f.wp = wp; f.bp = bp;
  for (int t = 0; t < 8; ++t) f.c2[t] = (float)(0.69314718055994530942 * 100.0 / (6.283185307179586476925 * (double)divisors8[t]));
  f.nq = (N + 31) / 32;
  const int ntile = (M + 31) / 32;
  const size_t lds = (size_t)2 * kGBuf * 4 + (size_t)ntile * 32 * 16;
  static relnet::PerDeviceOnce attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)relation_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const unsigned grid = (unsigned)(f.nq * ((B + 7) / 8) * 8);
  relation_fused_kernel<<<grid, 1024, lds, (hipStream_t)stream>>>(f);
  return check_launch("relnet_relation_attention_fused");
}
