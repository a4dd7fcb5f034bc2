// Learned duplicate removal (learn-NMS) head, test-time path (reference:
// relation_rcnn/operator_py/learn_nms.py:238-401 LearnNmsOperator.forward and its nd helpers;
// merge over thresholds: symbols/..._learn_nms.py:553-560).  The reference synchronises with the
// host twice inside the operator (class filtering via .asnumpy(), learn_nms.py:296-303,373-377);
// here the valid-class rule is evaluated on device and invalid classes simply produce zeros.
//
//   lnms_prepare_kernel   softmax over classes + refine_bbox_nd (:175-217) + clip, float32
//   lnms_sort_kernel      per (image, class): descending sort of the N scores, first_n ranks,
//                         gathers sorted_score / sorted_bbox (:289-308)
//   lnms_embed_kernel     X[b,c,r,:] = roi_feat_embedding[idx[r,c]] + rank_feat[r]  (:335-344)
//   lnms_score_kernel     Z = relu(X + attention), logit FC 128 -> T, sigmoid, x sorted_score,
//                         merge over thresholds, compaction of score > thresh (:349-381, tester.py:231-242)
// The 16-head attention in between (nms_attention_nd, :45-127) runs on the relation kernels with
// (image, class) pairs as the batch dimension and the 8-wide per-head value padded to the 64-wide
// tile (zero rows in the packed linear_out weight).
#include "common.h"

namespace relnet {

struct PrepArgs {
  const float* cls_score; long cs_ld;   // [B*N, C]
  const float* bbox_pred; long bp_ld;   // [B*N, 4*num_reg], class-agnostic fg deltas at +delta_off
  const float* rois;                    // [B*N, 5]
  const float* im_info;                 // [B, 3]
  float* prob;                          // [B*N, C-1]  (background dropped)
  float* boxes;                         // [B*N, 4]    refined + clipped
  int C, N, delta_off;
  float m0, m1, m2, m3, s0, s1, s2, s3; // bbox means / stds (0/1 when the graph passes None)
  const int* n_valid;                   // optional [B]: rows past n_valid[b] are padding -> probability 0 (they sort last)
};

#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void lnms_prepare_kernel(PrepArgs g) {
  const int r = blockIdx.x, lane = threadIdx.x;
  if (g.n_valid && (r % g.N) >= g.n_valid[r / g.N]) {
    for (int c = lane + 1; c < g.C; c += 64) g.prob[(long)r * (g.C - 1) + c - 1] = 0.f;
    if (lane == 0) *(float4*)(g.boxes + (long)r * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    return;
  }
  const float* z = g.cls_score + (long)r * g.cs_ld;
  float m = -INFINITY;
  for (int c = lane; c < g.C; c += 64) m = fmaxf(m, z[c]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float s = 0.f;
  for (int c = lane; c < g.C; c += 64) s += expf(z[c] - m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  for (int c = lane + 1; c < g.C; c += 64) g.prob[(long)r * (g.C - 1) + c - 1] = expf(z[c] - m) / s;
  if (lane == 0) {
    const float* roi = g.rois + (long)r * 5;
    const float* d = g.bbox_pred + (long)r * g.bp_ld + g.delta_off;
    const float* info = g.im_info + (long)(r / g.N) * 3;
    const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
    const float w = x2 - x1 + 1.f, h = y2 - y1 + 1.f;
    const float cx = 0.5f * (x1 + x2), cy = 0.5f * (y1 + y2);
    const float dx = d[0] * g.s0 + g.m0, dy = d[1] * g.s1 + g.m1, dw = d[2] * g.s2 + g.m2, dh = d[3] * g.s3 + g.m3;
    const float rcx = cx + w * dx, rcy = cy + h * dy;
    const float rw = w * (float)exp((double)dw), rh = h * (float)exp((double)dh);
    const float wo = 0.5f * (rw - 1.f), ho = 0.5f * (rh - 1.f);
    const float mx = info[1] - 1.f, my = info[0] - 1.f;
    float o[4] = {rcx - wo, rcy - ho, rcx + wo, rcy + ho};
    o[0] = fmaxf(fminf(o[0], mx), 0.f); o[1] = fmaxf(fminf(o[1], my), 0.f);
    o[2] = fmaxf(fminf(o[2], mx), 0.f); o[3] = fmaxf(fminf(o[3], my), 0.f);
    *(float4*)(g.boxes + (long)r * 4) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
#pragma clang fp contract(fast)

__device__ __forceinline__ unsigned int fkey32(float f) {
  const unsigned int u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct SortArgs {
  const float* prob;      // [B, N, NC]
  const float* boxes;     // [B, N, 4]
  int* rank_idx;          // [B, NC, F]
  float* sorted_score;    // [B, F, NC]
  float* sorted_bbox;     // [B, F, NC, 4]
  float* class_boxes;     // [B, NC, F, 4]   (class-major copy: "images" of the geometry kernel)
  float* class_max;       // [B, NC]
  int N, NC, F;
};

// One workgroup per (class, image); N <= 1024.  Ties: smaller roi index first.
__global__ __launch_bounds__(256) void lnms_sort_kernel(SortArgs g) {
  __shared__ unsigned long long keys[1024];
  const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  int np2 = 1;
  while (np2 < g.N) np2 <<= 1;
  for (int i = tid; i < np2; i += 256) {
    unsigned long long k = 0ull;
    if (i < g.N) k = ((unsigned long long)fkey32(g.prob[((long)b * g.N + i) * g.NC + c]) << 16) | (unsigned)(0xffff - i);
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < np2; i += 256) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], d = keys[ixj];
          const bool desc = (i & k) == 0;
          if (desc ? (a < d) : (a > d)) { keys[i] = d; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  for (int r = tid; r < g.F; r += 256) {
    const int idx = 0xffff - (int)(keys[r] & 0xffffu);
    const float s = g.prob[((long)b * g.N + idx) * g.NC + c];
    const float4 bx = *(const float4*)(g.boxes + ((long)b * g.N + idx) * 4);
    g.rank_idx[((long)b * g.NC + c) * g.F + r] = idx;
    g.sorted_score[((long)b * g.F + r) * g.NC + c] = s;
    *(float4*)(g.sorted_bbox + (((long)b * g.F + r) * g.NC + c) * 4) = bx;
    *(float4*)(g.class_boxes + (((long)b * g.NC + c) * g.F + r) * 4) = bx;
    if (r == 0) g.class_max[(long)b * g.NC + c] = s;
  }
}

struct EmbedArgs {
  const void* roi_emb;    // [B, N, D]  T
  const float* rank_feat; // [F, D] fp32
  const int* rank_idx;    // [B, NC, F]
  void* x;                // [B, NC, F, D] T
  int N, NC, F, D, bf16;
};

__global__ __launch_bounds__(256) void lnms_embed_kernel(EmbedArgs g) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;        // one thread per 8 features
  const int d8 = g.D >> 3;
  const int cg = (int)(t % d8);
  const long row = t / d8;                                   // (b, c, r) flattened
  const int b = blockIdx.y;
  if (row >= (long)g.NC * g.F) return;
  const int r = (int)(row % g.F);
  const int idx = g.rank_idx[(long)b * g.NC * g.F + row];
  const float* rf = g.rank_feat + (long)r * g.D + cg * 8;
  float v[8];
  if (g.bf16) {
    const uint4 e = *(const uint4*)((const unsigned short*)g.roi_emb + ((long)b * g.N + idx) * g.D + cg * 8);
    const unsigned int w4[4] = {e.x, e.y, e.z, e.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) { v[2 * k] = bf2f(w4[k] & 0xffff) + rf[2 * k]; v[2 * k + 1] = bf2f(w4[k] >> 16) + rf[2 * k + 1]; }
    *(uint4*)((unsigned short*)g.x + ((long)b * g.NC * g.F + row) * g.D + cg * 8) =
        make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
  } else {
    const float* e = (const float*)g.roi_emb + ((long)b * g.N + idx) * g.D + cg * 8;
    float* o = (float*)g.x + ((long)b * g.NC * g.F + row) * g.D + cg * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = e[k] + rf[k];
  }
}

struct ScoreArgs {
  const void* x;            // [B, NC, F, D] T
  const void* att;          // [B*NC, F, H*att_hstride] T  (padded per-head layout)
  const float* w_logit;     // [T, D]
  const float* b_logit;     // [T]
  const float* sorted_score;// [B, F, NC]
  const float* sorted_bbox; // [B, F, NC, 4]
  const float* class_max;   // [B, NC]
  float* multi;             // [B, F, NC, T]
  float* final_score;       // [B, F, NC]
  double* dets;             // [B, NC, F, 5] compacted score > thresh (x1,y1,x2,y2,score) or nullptr
  int* counts;              // [B, NC]
  const float* im_info;     // [B, 3] (boxes are divided by the image scale in dets)
  int NC, F, D, T, H, dv, att_hstride, bf16, merge;
  float class_thresh, score_thresh;
};

template <int D, int TT>
__global__ __launch_bounds__(64) void lnms_score_kernel(ScoreArgs g) {
  typedef const float __attribute__((address_space(4))) * cfloat_p;
  const int c = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
  // valid-class rule (learn_nms.py:293-302): max score of the class >= min(class_thresh, global max)
  float gm = -INFINITY;
  for (int k = lane; k < g.NC; k += 64) gm = fmaxf(gm, g.class_max[(long)b * g.NC + k]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) gm = fmaxf(gm, __shfl_xor(gm, o));
  const bool valid = g.class_max[(long)b * g.NC + c] >= fminf(g.class_thresh, gm);
  cfloat_p wl = (cfloat_p)(unsigned long long)g.w_logit;
  cfloat_p bl = (cfloat_p)(unsigned long long)g.b_logit;
  const float inv_scale = 1.0f / g.im_info[b * 3 + 2];
  int kept_before = 0;
  for (int r0 = 0; r0 < g.F; r0 += 64) {
    const int r = r0 + lane;
    float fin = 0.f;
    float ms[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) ms[t] = 0.f;
    if (r < g.F && valid) {
      const long xrow = (((long)b * g.NC + c) * g.F + r);
      float acc[TT];
#pragma unroll
      for (int t = 0; t < TT; ++t) acc[t] = bl[t];
      for (int d = 0; d < D; ++d) {
        const int h = d / g.dv, o = d - h * g.dv;
        float xv, av;
        if (g.bf16) {
          xv = bf2f(((const unsigned short*)g.x)[xrow * D + d]);
          av = bf2f(((const unsigned short*)g.att)[xrow * (long)(g.H * g.att_hstride) + h * g.att_hstride + o]);
        } else {
          xv = ((const float*)g.x)[xrow * D + d];
          av = ((const float*)g.att)[xrow * (long)(g.H * g.att_hstride) + h * g.att_hstride + o];
        }
        const float zv = fmaxf(xv + av, 0.f);
#pragma unroll
        for (int t = 0; t < TT; ++t) acc[t] = fmaf(zv, wl[t * D + d], acc[t]);
      }
      const float ss = g.sorted_score[((long)b * g.F + r) * g.NC + c];
      float sum = 0.f, mxv = -INFINITY;
#pragma unroll
      for (int t = 0; t < TT; ++t) {
        const float sg = 1.0f / (1.0f + expf(-acc[t]));
        ms[t] = ss * sg;
        sum += ms[t];
        mxv = fmaxf(mxv, ms[t]);
      }
      fin = g.merge == -1 ? sum / (float)TT : (g.merge == -2 ? mxv : ms[g.merge < TT ? (g.merge < 0 ? 0 : g.merge) : 0]);
    }
    if (r < g.F) {
      float* mo = g.multi + (((long)b * g.F + r) * g.NC + c) * TT;
#pragma unroll
      for (int t = 0; t < TT; ++t) mo[t] = ms[t];
      g.final_score[((long)b * g.F + r) * g.NC + c] = fin;
    }
    if (g.dets) {
      const bool keep = (r < g.F) && (fin > g.score_thresh);
      const unsigned long long bal = __ballot(keep);
      if (keep) {
        const int pos = kept_before + __builtin_popcountll(bal & ((1ull << lane) - 1ull));
        const float4 bx = *(const float4*)(g.sorted_bbox + (((long)b * g.F + r) * g.NC + c) * 4);
        double* o = g.dets + ((((long)b * g.NC + c) * g.F) + pos) * 5;
        o[0] = (double)(bx.x * inv_scale); o[1] = (double)(bx.y * inv_scale);
        o[2] = (double)(bx.z * inv_scale); o[3] = (double)(bx.w * inv_scale); o[4] = (double)fin;
      }
      kept_before += __builtin_popcountll(bal);
    }
  }
  if (g.counts && lane == 0) g.counts[(long)b * g.NC + c] = kept_before;
}

}  // namespace relnet

using namespace relnet;

extern "C" int relnet_lnms_prepare_ex(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld,
                                      const float* rois, const float* im_info, float* prob, float* boxes, int B,
                                      int N, int C, int delta_off, const float* means4, const float* stds4,
                                      const int* n_valid, void* stream) {
  RELNET_REQUIRE(cls_score && bbox_pred && rois && im_info && prob && boxes, "relnet_lnms_prepare: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && C > 1, "relnet_lnms_prepare: bad shape");
  PrepArgs g{cls_score, cs_ld, bbox_pred, bp_ld, rois, im_info, prob, boxes, C, N, delta_off,
             means4 ? means4[0] : 0.f, means4 ? means4[1] : 0.f, means4 ? means4[2] : 0.f, means4 ? means4[3] : 0.f,
             stds4 ? stds4[0] : 1.f, stds4 ? stds4[1] : 1.f, stds4 ? stds4[2] : 1.f, stds4 ? stds4[3] : 1.f, n_valid};
  lnms_prepare_kernel<<<B * N, 64, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_lnms_prepare");
}

extern "C" int relnet_lnms_prepare(const float* cls_score, long cs_ld, const float* bbox_pred, long bp_ld,
                                   const float* rois, const float* im_info, float* prob, float* boxes, int B,
                                   int N, int C, int delta_off, const float* means4, const float* stds4,
                                   void* stream) {
  return relnet_lnms_prepare_ex(cls_score, cs_ld, bbox_pred, bp_ld, rois, im_info, prob, boxes, B, N, C, delta_off, means4,
                                stds4, nullptr, stream);
}

extern "C" int relnet_lnms_sort(const float* prob, const float* boxes, int* rank_idx, float* sorted_score,
                                float* sorted_bbox, float* class_boxes, float* class_max, int B, int N, int NC,
                                int first_n, void* stream) {
  RELNET_REQUIRE(prob && boxes && rank_idx && sorted_score && sorted_bbox && class_boxes && class_max, "relnet_lnms_sort: null operand");
  RELNET_REQUIRE(B > 0 && N > 0 && N <= 1024 && NC > 0 && first_n > 0 && first_n <= N, "relnet_lnms_sort: need first_n <= N <= 1024");
  SortArgs g{prob, boxes, rank_idx, sorted_score, sorted_bbox, class_boxes, class_max, N, NC, first_n};
  lnms_sort_kernel<<<dim3(NC, B), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_lnms_sort");
}

extern "C" int relnet_lnms_embed(const void* roi_emb, const float* rank_feat, const int* rank_idx, void* x, int B,
                                 int N, int NC, int first_n, int D, int dtype, void* stream) {
  RELNET_REQUIRE(roi_emb && rank_feat && rank_idx && x, "relnet_lnms_embed: null operand");
  RELNET_REQUIRE(D % 8 == 0 && B > 0, "relnet_lnms_embed: D %% 8 required");
  EmbedArgs g{roi_emb, rank_feat, rank_idx, x, N, NC, first_n, D, dtype == 1};
  const long threads = (long)NC * first_n * (D / 8);
  lnms_embed_kernel<<<dim3((unsigned)((threads + 255) / 256), B), 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_lnms_embed");
}

extern "C" int relnet_lnms_score(const void* x, const void* att, const float* w_logit, const float* b_logit,
                                 const float* sorted_score, const float* sorted_bbox, const float* class_max,
                                 const float* im_info, float* multi, float* final_score, double* dets, int* counts,
                                 int B, int NC, int first_n, int D, int T, int H, int dv, int att_hstride,
                                 int merge, float class_thresh, float score_thresh, int dtype, void* stream) {
  RELNET_REQUIRE(x && att && w_logit && b_logit && sorted_score && sorted_bbox && class_max && im_info && multi && final_score,
                 "relnet_lnms_score: null operand");
  RELNET_REQUIRE(D == 128 && T == 5 && H * dv == D, "relnet_lnms_score: specialised to D=128, 5 thresholds (D=%d T=%d)", D, T);
  ScoreArgs g{x, att, w_logit, b_logit, sorted_score, sorted_bbox, class_max, multi, final_score, dets, counts, im_info,
              NC, first_n, D, T, H, dv, att_hstride, dtype == 1, merge, class_thresh, score_thresh};
  lnms_score_kernel<128, 5><<<dim3(NC, B), 64, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_lnms_score");
}
