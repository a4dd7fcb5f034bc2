// Element-wise pieces of the learn-NMS head's TRAIN branch (symbols/resnet_v1_101_rcnn_learn_nms_1024_attention_1024_pairwise_
// position_multi_head_16.py:424-551) and of its adjoint that the graph spells as chains of small tensor operators (slice + add + relu;
// sigmoid + transpose + multiply; the softmax Jacobian; take's adjoint ...).  Round 6: each chain is ONE kernel here -- at one image per
// GPU (the reference's protocol) the step is launch bound and these chains were ~50 of its ~450 launches.
//   relnet_lnms_pad_params     trained [128,128] / [T,128] matrices into the zero-padded operands of the 64-wide MFMA tiles
//   relnet_lnms_residual_relu  all_feat = relu(x + linear_out)                    (:489-491; 8 real of 64 columns per head)
//   relnet_lnms_cond_multi     conditional prob = sigmoid(logit), transposed to [B,F,C,T]; multi score = sorted score x cond  (:497-505)
//   relnet_lnms_cond_bwd       adjoint of the two lines above -> d sorted score, d logit (bf16, padded to 64 columns)
//   relnet_lnms_take_bwd       adjoint of take(roi_feat_embedding, rank indices)  (:447-452): gathered per roi over the classes that rank it, bf16 out
//   relnet_lnms_softmax_bwd    adjoint of softmax + slice_axis(begin=1) (:430-433), accumulated into the detector's d cls_score
//   relnet_lnms_gather_bias    the per-(image, class) geometry bias gathered from one per-image table by the class's ranks
#include "common.h"

namespace relnet {

__global__ __launch_bounds__(256) void lnms_pad_params_kernel(const unsigned short* wo, const float* bo, const unsigned short* wl, const float* bl,
                                                              unsigned short* wout_pad, float* bout_pad, unsigned short* wl_pad, float* bl_pad, int T) {
  // wo [16 heads x 8][128] -> rows h*64 + j of wout_pad [1024][128]; wl [T][128] -> rows 0..T-1 of wl_pad [64][128] (16-byte chunks)
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int nwo = 128 * 16, nwl = T * 16;
  if (i < nwo) {
    const int r = i >> 4, ch = i & 15;
    *((uint4*)(wout_pad + (long)((r >> 3) * 64 + (r & 7)) * 128) + ch) = *((const uint4*)(wo + (long)r * 128) + ch);
  } else if (i < nwo + nwl) {
    const int k = i - nwo;
    *((uint4*)wl_pad + k) = *((const uint4*)wl + k);
  } else if (i < nwo + nwl + 128) {
    const int r = i - nwo - nwl;
    bout_pad[(r >> 3) * 64 + (r & 7)] = bo[r];
  } else if (i < nwo + nwl + 128 + T) {
    const int r = i - nwo - nwl - 128;
    bl_pad[r] = bl[r];
  }
}

// out[r][h*8 + j] = relu(x[r][h*8 + j] + att[r][h*64 + j]); one thread = one (row, head): 16 bytes of each operand
__global__ __launch_bounds__(256) void lnms_residual_relu_kernel(const unsigned short* att, const unsigned short* x, unsigned short* out, long rows) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * 16) return;
  const long r = i >> 4; const int h = (int)(i & 15);
  const uint4 a = *(const uint4*)(att + r * 1024 + h * 64);
  const uint4 b = *(const uint4*)(x + r * 128 + h * 8);
  const unsigned int ua[4] = {a.x, a.y, a.z, a.w}, ub[4] = {b.x, b.y, b.z, b.w};
  unsigned int o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    // (the sum is rounded to bf16 BEFORE the ReLU, as the graph's broadcast_add + Activation do)
    const float lo = bf2f(f2bf(bf2f(ua[k] & 0xffff) + bf2f(ub[k] & 0xffff))), hi = bf2f(f2bf(bf2f(ua[k] >> 16) + bf2f(ub[k] >> 16)));
    o[k] = pack_bf16x2(fmaxf(lo, 0.f), fmaxf(hi, 0.f));
  }
  *(uint4*)(out + r * 128 + h * 8) = make_uint4(o[0], o[1], o[2], o[3]);
}

// thread = (b, f, c): row (b*C + c)*F + f of the logits
__global__ __launch_bounds__(256) void lnms_cond_multi_kernel(const float* logit, long ld, const float* score, float* cond, float* multi,
                                                              int B, int C, int F, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * F * C) return;
  const int c = (int)(i % C); const int f = (int)((i / C) % F); const long b = i / ((long)C * F);
  const float* lp = logit + ((b * C + c) * F + f) * ld;
  const float s = score[i];
  for (int t = 0; t < T; ++t) {
    const float p = 1.f / (1.f + __expf(-lp[t]));
    cond[i * T + t] = p;
    multi[i * T + t] = s * p;
  }
}

__global__ __launch_bounds__(256) void lnms_cond_bwd_kernel(const float* d_multi, const float* cond, const float* score, float* d_sorted,
                                                            unsigned short* d_logit, int B, int C, int F, int T) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * F * C) return;
  const int c = (int)(i % C); const int f = (int)((i / C) % F); const long b = i / ((long)C * F);
  const float s = score[i];
  float ds = 0.f;
  unsigned int row[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t < T) {
      const float dm = d_multi[i * T + t], p = cond[i * T + t];
      ds += dm * p;
      row[t] = f2bf(dm * s * p * (1.f - p));
    }
  }
  d_sorted[i] = ds;
  uint4* dp = (uint4*)(d_logit + ((b * C + c) * F + f) * 64);
  dp[0] = make_uint4(row[0] | (row[1] << 16), row[2] | (row[3] << 16), row[4] | (row[5] << 16), row[6] | (row[7] << 16));
#pragma unroll
  for (int q = 1; q < 8; ++q) dp[q] = make_uint4(0, 0, 0, 0);
}

// take's adjoint as a GATHER: d_emb[b N + n][:] = sum over the classes c that rank roi n (at position p) of d_x[b][c][p][:], written once as bf16.
// (First form, round 6: thread = 4 columns of one (b, c, f) row, fp32 atomics into a zeroed table -- a roi is ranked by up to C classes, the adds of one row
//  contend: 162 us at 8 images, plus the table's fill and its cast.)  One workgroup = 16 rois of one image: the image's C F ranks are scanned once for the
// positions of those rois (LDS table pos[C][16], -1 = not ranked), then thread (roi, 8 columns) adds the hits in fp32.
__global__ __launch_bounds__(256) void lnms_take_bwd_gather_kernel(const unsigned short* d_x, const int* rank_idx, unsigned short* d_emb, int N, int C, int F) {
  __shared__ short pos[128 * 16];                 // [C <= 128][16]
  const int chunks = (N + 15) / 16;
  const int b = blockIdx.x / chunks, n0 = (blockIdx.x % chunks) * 16;
  for (int i = threadIdx.x; i < C * 16; i += 256) pos[i] = -1;
  __syncthreads();
  const int* rk = rank_idx + (long)b * C * F;
  for (int i = threadIdx.x; i < C * F; i += 256) {
    const int r = rk[i] - n0;
    if (r >= 0 && r < 16) pos[(i / F) * 16 + r] = (short)(i % F);
  }
  __syncthreads();
  const int r = threadIdx.x >> 4, col = (threadIdx.x & 15) * 8;       // 16 rois x 16 column groups of 8
  const int n = n0 + r;
  if (n >= N) return;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const unsigned short* base = d_x + (long)b * C * F * 128 + col;
  for (int c = 0; c < C; ++c) {
    const int p = pos[c * 16 + r];
    if (p < 0) continue;
    const uint4 v = *(const uint4*)(base + ((long)c * F + p) * 128);
    const unsigned int u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(u[e] << 16); acc[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u); }
  }
  *(uint4*)(d_emb + ((long)b * N + n) * 128 + col) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
}

// one wavefront per (b, n): prob = softmax(cls_score)[1:], so with inner = sum_c prob_c d_prob_c and p_bg = 1 - sum_c prob_c:
//   d cls_score[0] += -p_bg inner;   d cls_score[1 + c] += prob_c (d_prob_c - inner)
__global__ __launch_bounds__(256) void lnms_softmax_bwd_kernel(const float* prob, const float* d_prob, float* d_cls, long ld_row, long ld_img,
                                                               int B, int N, int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)B * N) return;
  const long b = row / N, n = row - b * N;
  const float* pp = prob + row * C;
  const float* dp = d_prob + row * C;
  float sp = 0.f, si = 0.f;
  for (int c = lane; c < C; c += 64) { const float p = pp[c]; sp += p; si += p * dp[c]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); si += __shfl_xor(si, o, 64); }
  float* out = d_cls + b * ld_img + n * ld_row;
  if (lane == 0) out[0] += -(1.f - sp) * si;
  for (int c = lane; c < C; c += 64) out[1 + c] += pp[c] * (dp[c] - si);
}

// The geometry bias of the learn-NMS head's relation module, per (image, class): its boxes are the image's class-agnostic boxes re-ordered by the class's
// ranks, so ln G of (rank f1, rank f2) is the entry (r1, r2) of ONE per-image table.  out[bc][h][f1][f2] = img[b][h][rank[bc][f1]][rank[bc][f2]] for f2 < F
// (the pad columns F .. Fpad stay unwritten, like geometry_bias_kernel leaves them).
// The table costs B N^2 pairs instead of B C F^2 (8 x 300^2 against 640 x 100^2: 9 x fewer sin / cos / log evaluations); same arithmetic on the same box
// pairs -> bit-identical to the direct evaluation.  Negative ranks (padding of a short proposal list) are not handled here: the caller keeps the direct form.
__global__ __launch_bounds__(256) void lnms_gather_bias_kernel(const float* img, const int* rank, float* out, int C, int N, int Npad, int F, int Fpad) {
  // one workgroup per (image-class bc, head h): the class's F ranks staged in LDS once; thread = one f2 column of two f1 rows per trip -- a wavefront's
  // gathered loads stay inside one 1.3 KB table row, its stores are one contiguous row segment (first form: thread = 4 f2 of one row, 223 us at 8 images; this: 155 us)
  __shared__ int rk[512];
  const long bc = blockIdx.x >> 4;
  const int h = blockIdx.x & 15;
  const long b = bc / C;
  for (int i = threadIdx.x; i < F; i += 256) rk[i] = rank[bc * F + i];
  __syncthreads();
  const int cols = (F + 63) & ~63;              // columns walked per row, a multiple of the wavefront
  const int rows_per_trip = 256 / cols > 0 ? 256 / cols : 1;
  const float* tab = img + (b * 16 + h) * (long)N * Npad;
  float* o = out + (bc * 16 + h) * (long)F * Fpad;
  if (cols <= 256) {
    const int f2 = threadIdx.x % cols, dr = threadIdx.x / cols;
    if (dr < rows_per_trip && f2 < F) {
      const int c2 = rk[f2];
      for (int f1 = dr; f1 < F; f1 += rows_per_trip) o[(long)f1 * Fpad + f2] = tab[(long)rk[f1] * Npad + c2];
    }
  } else {
    for (int f1 = 0; f1 < F; ++f1)
      for (int f2 = threadIdx.x; f2 < F; f2 += 256) o[(long)f1 * Fpad + f2] = tab[(long)rk[f1] * Npad + rk[f2]];
  }
}

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_lnms_gather_bias(const float* img, const int* rank_idx, float* out, int B, int C, int N, int Npad, int F, int Fpad, void* stream) {
  RELNET_REQUIRE(img && rank_idx && out && B > 0 && C > 0 && N > 0 && F > 0 && Npad >= N && Fpad >= F && Fpad % 4 == 0, "relnet_lnms_gather_bias: bad arguments");
  RELNET_REQUIRE((((uintptr_t)out) & 15) == 0, "relnet_lnms_gather_bias: out must be 16-byte aligned");
  RELNET_REQUIRE(F <= 512, "relnet_lnms_gather_bias: first_n %d > 512", F);
  lnms_gather_bias_kernel<<<(unsigned)((long)B * C * 16), 256, 0, (hipStream_t)stream>>>(img, rank_idx, out, C, N, Npad, F, Fpad);
  return check_launch("relnet_lnms_gather_bias");
}

extern "C" int relnet_lnms_pad_params(const void* wo, const float* bo, const void* wl, const float* bl, void* wout_pad, float* bout_pad,
                                      void* wl_pad, float* bl_pad, int T, void* stream) {
  RELNET_REQUIRE(wo && bo && wl && bl && wout_pad && bout_pad && wl_pad && bl_pad && T > 0 && T <= 64, "relnet_lnms_pad_params: bad arguments");
  const int total = 128 * 16 + T * 16 + 128 + T;
  lnms_pad_params_kernel<<<(total + 255) / 256, 256, 0, (hipStream_t)stream>>>((const unsigned short*)wo, bo, (const unsigned short*)wl, bl,
                                                                               (unsigned short*)wout_pad, bout_pad, (unsigned short*)wl_pad, bl_pad, T);
  return check_launch("relnet_lnms_pad_params");
}

extern "C" int relnet_lnms_residual_relu(const void* att, const void* x, void* out, long rows, void* stream) {
  RELNET_REQUIRE(att && x && out && rows > 0, "relnet_lnms_residual_relu: bad arguments");
  RELNET_REQUIRE((((uintptr_t)att | (uintptr_t)x | (uintptr_t)out) & 15) == 0, "relnet_lnms_residual_relu: operands must be 16-byte aligned");
  lnms_residual_relu_kernel<<<(unsigned)((rows * 16 + 255) / 256), 256, 0, (hipStream_t)stream>>>((const unsigned short*)att, (const unsigned short*)x,
                                                                                                 (unsigned short*)out, rows);
  return check_launch("relnet_lnms_residual_relu");
}

extern "C" int relnet_lnms_cond_multi(const float* logit, long ld, const float* sorted_score, float* cond, float* multi, int B, int C, int F, int T,
                                      void* stream) {
  RELNET_REQUIRE(logit && sorted_score && cond && multi && B > 0 && C > 0 && F > 0 && T > 0 && ld >= T, "relnet_lnms_cond_multi: bad arguments");
  const long total = (long)B * F * C;
  lnms_cond_multi_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(logit, ld, sorted_score, cond, multi, B, C, F, T);
  return check_launch("relnet_lnms_cond_multi");
}

extern "C" int relnet_lnms_cond_bwd(const float* d_multi, const float* cond, const float* sorted_score, float* d_sorted, void* d_logit, int B, int C,
                                    int F, int T, void* stream) {
  RELNET_REQUIRE(d_multi && cond && sorted_score && d_sorted && d_logit && B > 0 && C > 0 && F > 0 && T > 0 && T <= 8, "relnet_lnms_cond_bwd: bad arguments (T <= 8)");
  RELNET_REQUIRE((((uintptr_t)d_logit) & 15) == 0, "relnet_lnms_cond_bwd: d_logit must be 16-byte aligned");
  const long total = (long)B * F * C;
  lnms_cond_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_multi, cond, sorted_score, d_sorted, (unsigned short*)d_logit, B, C, F, T);
  return check_launch("relnet_lnms_cond_bwd");
}

extern "C" int relnet_lnms_take_bwd(const void* d_x, const int* rank_idx, void* d_emb, int B, int N, int C, int F, void* stream) {
  RELNET_REQUIRE(d_x && rank_idx && d_emb && B > 0 && N > 0 && C > 0 && C <= 128 && F > 0 && F < 32768, "relnet_lnms_take_bwd: bad arguments (C <= 128)");
  RELNET_REQUIRE((((uintptr_t)d_x | (uintptr_t)d_emb) & 15) == 0, "relnet_lnms_take_bwd: operands must be 16-byte aligned");
  lnms_take_bwd_gather_kernel<<<(unsigned)(B * ((N + 15) / 16)), 256, 0, (hipStream_t)stream>>>((const unsigned short*)d_x, rank_idx, (unsigned short*)d_emb, N, C, F);
  return check_launch("relnet_lnms_take_bwd");
}

extern "C" int relnet_lnms_softmax_bwd(const float* prob, const float* d_prob, float* d_cls, long ld_row, long ld_img, int B, int N, int C, void* stream) {
  RELNET_REQUIRE(prob && d_prob && d_cls && B > 0 && N > 0 && C > 0 && ld_row >= C + 1, "relnet_lnms_softmax_bwd: bad arguments");
  const long rows = (long)B * N;
  lnms_softmax_bwd_kernel<<<(unsigned)((rows + 3) / 4), 256, 0, (hipStream_t)stream>>>(prob, d_prob, d_cls, ld_row, ld_img, B, N, C);
  return check_launch("relnet_lnms_softmax_bwd");
}
