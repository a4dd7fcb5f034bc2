// Elementwise pieces of the training step:
//  * relnet_relu_bwd    gradient through mx.symbol.Activation(act_type='relu') (and through the fused
//                       conv + bias + [residual] + ReLU epilogues of the forward kernels): dx = dy * (y > 0),
//                       optionally + an accumulated second gradient (the bottleneck's shortcut branch).
//  * relnet_colsum_add  bias gradients: column sums of an upstream gradient accumulated into the flat gradient buffer.
//  * relnet_sgd_update  mx.optimizer.SGD as configured by relation_rcnn/train_end2end.py:163-168
//                       (momentum 0.9, wd 5e-4, rescale_grad 1.0, no gradient clipping):
//                         mom = momentum * mom - lr * (rescale * grad + wd * w);  w += mom
//                       on fp32 master weights, optionally refreshing a bf16 copy for the MFMA kernels.
#include "common.h"

namespace relnet {
enum { RELNET_F32 = 0, RELNET_BF16 = 1 };

template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* dy, const T* y, const T* add, T* dx, long n) {
  constexpr int V = 16 / sizeof(T);
  const long nv = n / V;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    const uint4 g = *((const uint4*)dy + i), o = *((const uint4*)y + i);
    uint4 a = make_uint4(0, 0, 0, 0);
    if (add) a = *((const uint4*)add + i);
    uint4 r;
    if constexpr (sizeof(T) == 4) {
      const float* fg = (const float*)&g; const float* fo = (const float*)&o; const float* fa = (const float*)&a;
      float* fr = (float*)&r;
#pragma unroll
      for (int k = 0; k < 4; ++k) fr[k] = (fo[k] > 0.f ? fg[k] : 0.f) + (add ? fa[k] : 0.f);
    } else {
      const unsigned int* ug = (const unsigned int*)&g; const unsigned int* uo = (const unsigned int*)&o;
      const unsigned int* ua = (const unsigned int*)&a; unsigned int* ur = (unsigned int*)&r;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float glo = __uint_as_float(ug[k] << 16), ghi = __uint_as_float(ug[k] & 0xffff0000u);
        const float olo = __uint_as_float(uo[k] << 16), ohi = __uint_as_float(uo[k] & 0xffff0000u);
        const float alo = add ? __uint_as_float(ua[k] << 16) : 0.f, ahi = add ? __uint_as_float(ua[k] & 0xffff0000u) : 0.f;
        ur[k] = pack_bf16x2((olo > 0.f ? glo : 0.f) + alo, (ohi > 0.f ? ghi : 0.f) + ahi);
      }
    }
    *((uint4*)dx + i) = r;
  }
  // tail (n not a multiple of the vector width)
  if (blockIdx.x == 0) {
    for (long i = nv * V + threadIdx.x; i < n; i += 256) {
      float g, o, a = 0.f;
      if constexpr (sizeof(T) == 4) { g = dy[i]; o = y[i]; if (add) a = add[i]; dx[i] = (o > 0.f ? g : 0.f) + a; }
      else { g = bf2f(dy[i]); o = bf2f(y[i]); if (add) a = bf2f(add[i]); dx[i] = f2bf((o > 0.f ? g : 0.f) + a); }
    }
  }
}

// out[c] += sum over rows of x[r, c]  (bias gradients: the column sum of an upstream gradient [rows, cols], bf16 or fp32, row stride ld,
// accumulated in fp32 into the flat gradient buffer).  Block = 64 columns x 4 row lanes over one chunk of rows; one fp32 atomic per
// column per block.  Replaces `.float().sum(0)` + `add_` (a conversion, a reduction and an accumulation launch per bias).
template <typename T>
__global__ __launch_bounds__(256) void colsum_add_kernel(const T* x, long ld, long rows, int cols, long rows_per_block, float* out) {
  __shared__ float part[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  float acc = 0.f;
  if (c < cols) {
    for (long r = r0 + rl; r < r1; r += 4) {
      if constexpr (sizeof(T) == 4) acc += x[r * ld + c];
      else acc += bf2f(x[r * ld + c]);
    }
  }
  part[rl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (rl == 0 && c < cols) atomicAdd(out + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// bf16 rows with cols % 8 == 0 and 16-byte aligned rows: 16-byte loads, thread = (8-column group, row lane); block = 32 groups x 8 row lanes
__global__ __launch_bounds__(256) void colsum_add_bf16x8_kernel(const unsigned short* x, long ld, long rows, int cols, long rows_per_block, float* out) {
  __shared__ float part[8][32][9];
  const int gq = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (blockIdx.x * 32 + gq) * 8;
  const long r0 = (long)blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    for (long r = r0 + rl; r < r1; r += 8) {
      const uint4 v = *(const uint4*)(x + r * ld + c);
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(w4[e] << 16); acc[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[rl][gq][e] = acc[e];
  __syncthreads();
  if (rl == 0 && c < cols) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += part[k][gq][e];
      atomicAdd(out + c + e, t);
    }
  }
}

// Grouped form (round 6): the bias gradients of a gradient bucket in ONE launch -- up to 16 (matrix, output) problems in the kernel argument
// (no table in memory: capture safe), every workgroup finds its problem by a scan of the block prefix and then runs the body of
// colsum_add_bf16x8_kernel.  A training step has 12 - 13 bias gradients (RPN head, conv_new_1, the FC layers, the relation modules' projections), all of
// the heads bucket: 12 launches of 17 - 29 us (latency bound reductions) become one.
struct ColsumProblem { const unsigned short* x; float* out; long ld, rows, rows_per_block; int cols, col_blocks, blk_start; };
struct ColsumGroup { ColsumProblem p[16]; int n; };

__global__ __launch_bounds__(256) void colsum_add_grouped_kernel(ColsumGroup g) {
  __shared__ float part[8][32][9];
  int pi = 0;
  for (int i = 1; i < g.n; ++i) if ((int)blockIdx.x >= g.p[i].blk_start) pi = i;
  const ColsumProblem& a = g.p[pi];
  const int rel = blockIdx.x - a.blk_start;
  const int bx = rel % a.col_blocks, by = rel / a.col_blocks;
  const int gq = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int c = (bx * 32 + gq) * 8;
  const long r0 = (long)by * a.rows_per_block, r1 = min(r0 + a.rows_per_block, a.rows);
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < a.cols) {
    for (long r = r0 + rl; r < r1; r += 8) {
      const uint4 v = *(const uint4*)(a.x + r * a.ld + c);
      const unsigned int w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { acc[2 * e] += __uint_as_float(w4[e] << 16); acc[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u); }
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) part[rl][gq][e] = acc[e];
  __syncthreads();
  if (rl == 0 && c < a.cols) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) t += part[k][gq][e];
      atomicAdd(a.out + c + e, t);
    }
  }
}

struct SgdArgs {
  float* w; float* mom; const float* grad; unsigned short* w_bf16;
  long n;
  float lr, momentum, wd, rescale;
};

__global__ __launch_bounds__(256) void sgd_update_kernel(SgdArgs g) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < g.n; i += (long)gridDim.x * 256) {
    const float w = g.w[i];
    const float m = g.momentum * g.mom[i] - g.lr * (g.rescale * g.grad[i] + g.wd * w);
    const float wn = w + m;
    g.mom[i] = m;
    g.w[i] = wn;
    if (g.w_bf16) g.w_bf16[i] = f2bf(wn);
  }
}

// grad[r, c] += scale[r]^2 * sum_s parts[s, r, c]: the split-K partial sums of a weight gradient, the frozen-BatchNorm
// factor of a folded convolution (w' = w s  =>  dL/dw' enters SGD as s^2 dL/dw') and the accumulation into the flat
// gradient buffer in ONE pass (instead of torch sum + mul + add_).
__global__ __launch_bounds__(256) void wgrad_accumulate_kernel(const float* parts, int splits, long per_split, int cols,
                                                               const float* row_scale, float* grad) {
  const long nv = per_split / 4;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    float4 acc = *((const float4*)parts + i);
    for (int s = 1; s < splits; ++s) {
      const float4 v = *((const float4*)(parts + (long)s * per_split) + i);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    float m = 1.f;
    if (row_scale) { const float sc = row_scale[(i * 4) / cols]; m = sc * sc; }     // cols % 4 == 0: one row per float4
    float4 g = *((float4*)grad + i);
    g.x += m * acc.x; g.y += m * acc.y; g.z += m * acc.z; g.w += m * acc.w;
    *((float4*)grad + i) = g;
  }
}

// ---------------------------------------------------------------------------------------
// relnet_weight_relayout: the data-gradient layouts of ALL weights of the step in one launch.
// The data gradient of y = conv(x, W) is a convolution of dy with the tap-flipped, (Cout, Cin)-transposed filter; round 2 / 3
// produced those copies layer by layer inside the backward pass (87 transposes + ~30 flip / copy launches per step, ~1.2 ms at
// 8 images).  The weights only change in the SGD kernel, so one grouped launch after it rewrites every copy:
//   dst[ci][(taps - 1 - tap) * dst_co + co] = src[co][tap * cin + ci]        (taps = 1: a plain transpose, dst_co >= cout zero padded
//   by the caller's initial memset -- the pad columns are never written)
// One workgroup = one 64 (co) x 64 (ci) tile of one tap of one problem, found by binary search in the tile prefix of the table.
// ---------------------------------------------------------------------------------------
struct RelayoutProblem {            // mirrors relnet_relayout_desc
  const unsigned short* src; unsigned short* dst;
  int cout, cin, taps, dst_ld;      // dst_ld = taps * dst_co (elements)
  int dst_co, tiles_co, tiles_ci, tile_start;
};

__global__ __launch_bounds__(256) void weight_relayout_kernel(const RelayoutProblem* tab, int n) {
  __shared__ unsigned short tile[64][66];
  int lo = 0, hi = n - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {                                 // last problem with tile_start <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile_start <= bid) lo = mid; else hi = mid - 1;
  }
  const RelayoutProblem p = tab[lo];
  int t = bid - p.tile_start;
  const int tci = t % p.tiles_ci; t /= p.tiles_ci;
  const int tco = t % p.tiles_co;
  const int tap = t / p.tiles_co;
  const int co0 = tco * 64, ci0 = tci * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long src_ld = (long)p.taps * p.cin;
#pragma unroll
  for (int r = ty; r < 64; r += 4) {
    const int co = co0 + r, ci = ci0 + tx;
    tile[r][tx] = (co < p.cout && ci < p.cin) ? p.src[(long)co * src_ld + (long)tap * p.cin + ci] : (unsigned short)0;
  }
  __syncthreads();
  const int tapf = p.taps - 1 - tap;
#pragma unroll
  for (int r = ty; r < 64; r += 4) {
    const int ci = ci0 + r, co = co0 + tx;
    if (ci < p.cin && co < p.cout) p.dst[(long)ci * p.dst_ld + (long)tapf * p.dst_co + co] = tile[tx][r];
  }
}


// ---------------------------------------------------------------------------------------
// relnet_weight_fragpack: the MFMA-fragment-order copies of the TRAINED weights the chain kernels read (csrc/bottleneck.hip), all
// layers in one launch per step -- the training-time twin of relnet_pack_w_frag / ops.pack_chain_w1, which inference runs once at
// load time.  Thread = one 16-byte fragment slot:
//   mode 0 (W3, expand product; relnet_pack_w_frag order)   block (n / 32, k / 16), lane l <- W[32 nb + (l & 31)][16 kb + 8 (l >> 5) + 0..7]
//   mode 1 (W1', the next unit's reduce; ops.pack_chain_w1)  block (n / 32, k / 16), lane l, slot t <- W[32 nb + (l & 31)][16 kb + 8 (t >> 2) + 4 (l >> 5) + (t & 3)]
//          (the contraction index in the order the accumulator registers of the expand product hold it)
// table: DEVICE array; block_start = exclusive prefix of ceil(slots / 256).
// ---------------------------------------------------------------------------------------
struct FragPackProblem {            // mirrors relnet_fragpack_desc
  const unsigned short* src; uint4* dst;
  long ldw;
  int N, K, mode, block_start;
};

__global__ __launch_bounds__(256) void weight_fragpack_kernel(const FragPackProblem* tab, int n) {
  int lo = 0, hi = n - 1;
  const int bid = blockIdx.x;
  while (lo < hi) {                                 // last problem with block_start <= bid
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].block_start <= bid) lo = mid; else hi = mid - 1;
  }
  const FragPackProblem p = tab[lo];
  const long t = (long)(bid - p.block_start) * 256 + threadIdx.x;
  const int kblocks = p.K / 16;
  const long total = (long)(p.N / 32) * kblocks * 64;
  if (t >= total) return;
  const int l = (int)(t & 63);
  const long blk = t >> 6;
  const int kb = (int)(blk % kblocks), nb = (int)(blk / kblocks);
  const unsigned short* row = p.src + (long)(nb * 32 + (l & 31)) * p.ldw + kb * 16;
  if (p.mode == 0) {
    p.dst[t] = *(const uint4*)(row + 8 * (l >> 5));
  } else {
    const uint2 a = *(const uint2*)(row + 4 * (l >> 5)), b = *(const uint2*)(row + 8 + 4 * (l >> 5));
    p.dst[t] = make_uint4(a.x, a.y, b.x, b.y);
  }
}

// ---------------------------------------------------------------------------------------
// relnet_relation_bwd_pack: the three fp32 gradients the attention backward kernels produce -- dQ [B][N][d], dK and dVW [B][M][d]
// (M <= N keys = the first M rows) -- rounded to bf16 into ONE row-major operand A3 [B][N][3 d] = (dQ | dK | dVW), rows >= M of the
// key blocks zero.  With it the projections' backward is one GEMM and one weight-gradient product:
//   dF = A3 . [Wq; Wk; Wout]  (K = 3 d; the residual gradient rides in the epilogue),   d[Wq; Wk; Wout] = A3^T F
// instead of two GEMMs, two products and ~10 elementwise launches (zero fill, two strided copies, conversions, adds).
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relation_bwd_pack_kernel(const float* dq, const float* dk, const float* dvw, unsigned short* out,
                                                                int N, int M, int d, long total8) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total8; i += (long)gridDim.x * 256) {
    const int per_row = 3 * d / 8;
    const long row = i / per_row;                     // b * N + n
    const int c = (int)(i - row * per_row) * 8;       // column in [0, 3 d)
    const int n = (int)(row % N);
    const long b = row / N;
    const int blk = c / d, cc = c - blk * d;
    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
    if (blk == 0) {
      const float* s = dq + row * d + cc;
      v0 = *(const float4*)s; v1 = *(const float4*)(s + 4);
    } else if (n < M) {
      const float* s = (blk == 1 ? dk : dvw) + (b * M + n) * (long)d + cc;
      v0 = *(const float4*)s; v1 = *(const float4*)(s + 4);
    }
    *(uint4*)(out + row * 3 * d + c) = make_uint4(pack_bf16x2(v0.x, v0.y), pack_bf16x2(v0.z, v0.w), pack_bf16x2(v1.x, v1.y), pack_bf16x2(v1.z, v1.w));
  }
}

// ---------------------------------------------------------------------------------------
// relnet_lnms_scatter_bwd: adjoint of the learn-NMS head's per-class sort + slice (symbols/..._learn_nms.py:438-446: argsort of the
// class scores, take of the first_n rows):  d_prob[b][rank_idx[b][c][f]][c] += d_sorted[b][f][c].  One thread per (b, c, f);
// replaces torch's index_put_(accumulate=True), which sorts its 64 000 indices first (~12 launches).  Entries with a negative
// rank (padding of a short proposal list) are skipped.  d_prob must be zeroed by the caller.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lnms_scatter_bwd_kernel(const float* d_sorted, const int* rank_idx, float* d_prob, int B, int N, int C, int F) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * C * F) return;
  const int f = (int)(i % F);
  const int c = (int)((i / F) % C);
  const long b = i / ((long)F * C);
  const int r = rank_idx[i];
  if (r < 0 || r >= N) return;
  atomicAdd(d_prob + (b * N + r) * C + c, d_sorted[(b * F + f) * C + c]);
}

// out[0] += scale * sum(x)  (mode 0)  or  the number of entries >= 0 (mode 1): the scalar metrics of a training step (loss values =
// MakeLoss outputs summed per image, the OHEM keep count) in one launch each instead of sum + div / ge + sum.  Up to 64 workgroups, one float
// atomic per workgroup into the slot the entry point has zeroed (a first version with ONE workgroup took 106 us on the 690 000 RPN loss terms of
// an 8-image step: 0.5 ms per step, profiles/r06_notes).
__global__ __launch_bounds__(256) void reduce_scalar_kernel(const float* x, long n, float scale, int mode, float* out) {
  __shared__ float part[4];
  float acc = 0.f;
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) acc += mode ? (x[i] >= 0.f ? 1.f : 0.f) : x[i];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (mode ? 1.f : scale) * (part[0] + part[1] + part[2] + part[3]));
}


// Adjoint of a stride-s 1x1 convolution's input sampling (x[:, ::s, ::s, :]): the low-resolution data gradient goes to the sampled pixels
// of a full-resolution map, every other pixel is zero -- one pass that also applies the ReLU mask of the map's producer when that map has no
// other consumer (zero fill + strided copy + relu_bwd otherwise: three passes over the full-resolution map).
template <typename T>
__global__ __launch_bounds__(256) void strided_scatter_kernel(const T* low, const T* mask, T* out, long nvec, int H, int W, int CV, int Ho, int Wo,
                                                              int stride) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long)gridDim.x * 256) {
    const int cv = (int)(i % CV);
    long p = i / CV;
    const int w = (int)(p % W); p /= W;
    const int h = (int)(p % H);
    const long b = p / H;
    uint4 r = make_uint4(0, 0, 0, 0);
    const int ho = h / stride, wo = w / stride;
    if (ho * stride == h && wo * stride == w && ho < Ho && wo < Wo) {
      r = *((const uint4*)low + ((b * Ho + ho) * Wo + wo) * CV + cv);
      if (mask) {
        const uint4 o = *((const uint4*)mask + i);
        const unsigned int* uo = (const unsigned int*)&o;
        unsigned int* ur = (unsigned int*)&r;
        if constexpr (sizeof(T) == 4) {
#pragma unroll
          for (int k = 0; k < 4; ++k) ur[k] = __uint_as_float(uo[k]) > 0.f ? ur[k] : 0u;
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const float olo = __uint_as_float(uo[k] << 16), ohi = __uint_as_float(uo[k] & 0xffff0000u);
            ur[k] = (olo > 0.f ? (ur[k] & 0xffffu) : 0u) | (ohi > 0.f ? (ur[k] & 0xffff0000u) : 0u);
          }
        }
      }
    }
    *((uint4*)out + i) = r;
  }
}
}  // namespace relnet

using namespace relnet;

extern "C" int relnet_reduce_scalar(const float* x, long n, float scale, int mode, float* out, void* stream) {
  RELNET_REQUIRE(x && out && n > 0 && (mode == 0 || mode == 1), "relnet_reduce_scalar: bad arguments");
  if (hipMemsetAsync(out, 0, sizeof(float), (hipStream_t)stream) != hipSuccess) return check_launch("relnet_reduce_scalar(memset)");
  long blocks = (n + 4095) / 4096;
  blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
  reduce_scalar_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, n, scale, mode, out);
  return check_launch("relnet_reduce_scalar");
}

extern "C" int relnet_relu_bwd(const void* dy, const void* y, const void* add, void* dx, long n, int dtype, void* stream) {
  RELNET_REQUIRE(dy && y && dx && n > 0, "relnet_relu_bwd: bad operand");
  RELNET_REQUIRE((((uintptr_t)dy | (uintptr_t)y | (uintptr_t)dx | (uintptr_t)add) & 15) == 0, "relnet_relu_bwd: operands must be 16-byte aligned");
  long blocks = (n / 8 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RELNET_F32) relu_bwd_kernel<float><<<(unsigned)blocks, 256, 0, s>>>((const float*)dy, (const float*)y, (const float*)add, (float*)dx, n);
  else if (dtype == RELNET_BF16) relu_bwd_kernel<unsigned short><<<(unsigned)blocks, 256, 0, s>>>((const unsigned short*)dy, (const unsigned short*)y, (const unsigned short*)add, (unsigned short*)dx, n);
  else RELNET_REQUIRE(false, "relnet_relu_bwd: unknown dtype %d", dtype);
  return check_launch("relnet_relu_bwd");
}

extern "C" int relnet_strided_scatter(const void* low, const void* mask, void* out, int B, int H, int W, int C, int Ho, int Wo, int stride, int dtype,
                                      void* stream) {
  RELNET_REQUIRE(low && out && B > 0 && H > 0 && W > 0 && C > 0 && stride >= 1, "relnet_strided_scatter: bad operand");
  RELNET_REQUIRE(Ho == (H - 1) / stride + 1 && Wo == (W - 1) / stride + 1, "relnet_strided_scatter: [%d, %d] is not the stride-%d sampling of [%d, %d]", Ho, Wo, stride, H, W);
  RELNET_REQUIRE(dtype == RELNET_F32 || dtype == RELNET_BF16, "relnet_strided_scatter: unknown dtype %d", dtype);
  const int V = dtype == RELNET_F32 ? 4 : 8;
  RELNET_REQUIRE(C % V == 0, "relnet_strided_scatter: channels (%d) must be a multiple of %d", C, V);
  RELNET_REQUIRE((((uintptr_t)low | (uintptr_t)mask | (uintptr_t)out) & 15) == 0, "relnet_strided_scatter: operands must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const long nvec = (long)B * H * W * (C / V);
  const long blocks = std::min<long>((nvec + 255) / 256, 65536);
  if (dtype == RELNET_F32) strided_scatter_kernel<float><<<(unsigned)blocks, 256, 0, s>>>((const float*)low, (const float*)mask, (float*)out, nvec, H, W, C / V, Ho, Wo, stride);
  else strided_scatter_kernel<unsigned short><<<(unsigned)blocks, 256, 0, s>>>((const unsigned short*)low, (const unsigned short*)mask, (unsigned short*)out, nvec, H, W, C / V, Ho, Wo, stride);
  return check_launch("relnet_strided_scatter");
}

extern "C" int relnet_colsum_add(const void* x, long ld, long rows, int cols, int dtype, float* out, void* stream) {
  RELNET_REQUIRE(x && out && rows > 0 && cols > 0 && ld >= cols, "relnet_colsum_add: bad operand");
  long nchunk = (rows + 511) / 512;
  nchunk = nchunk < 1 ? 1 : (nchunk > 128 ? 128 : nchunk);
  const long rpb = (rows + nchunk - 1) / nchunk;
  dim3 grid((unsigned)((cols + 63) / 64), (unsigned)((rows + rpb - 1) / rpb));
  hipStream_t s = (hipStream_t)stream;
  if (dtype == RELNET_BF16 && cols % 8 == 0 && ld % 8 == 0 && ((uintptr_t)x & 15) == 0) {
    dim3 g8((unsigned)((cols / 8 + 31) / 32), grid.y);
    colsum_add_bf16x8_kernel<<<g8, 256, 0, s>>>((const unsigned short*)x, ld, rows, cols, rpb, out);
    return check_launch("relnet_colsum_add");
  }
  if (dtype == RELNET_F32) colsum_add_kernel<float><<<grid, 256, 0, s>>>((const float*)x, ld, rows, cols, rpb, out);
  else if (dtype == RELNET_BF16) colsum_add_kernel<unsigned short><<<grid, 256, 0, s>>>((const unsigned short*)x, ld, rows, cols, rpb, out);
  else RELNET_REQUIRE(false, "relnet_colsum_add: unknown dtype %d", dtype);
  return check_launch("relnet_colsum_add");
}

// n <= 16 problems: xs[i] bf16 [rows[i]][cols[i]] with row pitch lds[i] (cols % 8 == 0, ld % 8 == 0, 16-byte aligned), outs[i] fp32 [cols[i]] += column sums
extern "C" int relnet_colsum_add_grouped(const void* const* xs, const long* lds, const long* rows, const int* cols, float* const* outs, int n, void* stream) {
  RELNET_REQUIRE(xs && lds && rows && cols && outs && n > 0 && n <= 16, "relnet_colsum_add_grouped: 1..16 problems, got %d", n);
  ColsumGroup g;
  g.n = n;
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    RELNET_REQUIRE(xs[i] && outs[i] && rows[i] > 0 && cols[i] > 0 && cols[i] % 8 == 0 && lds[i] % 8 == 0 && lds[i] >= cols[i] && (((uintptr_t)xs[i]) & 15) == 0,
                   "relnet_colsum_add_grouped: problem %d needs bf16 rows of 8 | cols, 8 | ld, 16-byte aligned", i);
    long nchunk = (rows[i] + 511) / 512;
    nchunk = nchunk < 1 ? 1 : (nchunk > 128 ? 128 : nchunk);
    ColsumProblem& p = g.p[i];
    p.x = (const unsigned short*)xs[i]; p.out = outs[i]; p.ld = lds[i]; p.rows = rows[i]; p.cols = cols[i];
    p.rows_per_block = (rows[i] + nchunk - 1) / nchunk;
    p.col_blocks = (cols[i] / 8 + 31) / 32;
    p.blk_start = blocks;
    blocks += p.col_blocks * (int)((rows[i] + p.rows_per_block - 1) / p.rows_per_block);
  }
  colsum_add_grouped_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_colsum_add_grouped");
}

extern "C" int relnet_sgd_update(float* w, float* mom, const float* grad, void* w_bf16, long n, float lr, float momentum,
                                 float wd, float rescale_grad, void* stream) {
  RELNET_REQUIRE(w && mom && grad && n > 0, "relnet_sgd_update: bad operand");
  SgdArgs g{w, mom, grad, (unsigned short*)w_bf16, n, lr, momentum, wd, rescale_grad};
  long blocks = (n + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  sgd_update_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(g);
  return check_launch("relnet_sgd_update");
}

extern "C" int relnet_wgrad_accumulate(const float* parts, int splits, long rows, int cols, const float* row_scale,
                                       float* grad, void* stream) {
  RELNET_REQUIRE(parts && grad && splits > 0 && rows > 0 && cols > 0, "relnet_wgrad_accumulate: bad operand");
  RELNET_REQUIRE(cols % 4 == 0 && (((uintptr_t)parts | (uintptr_t)grad) & 15) == 0,
                 "relnet_wgrad_accumulate: cols %% 4 and 16-byte alignment required (cols=%d)", cols);
  const long per_split = rows * cols;
  long blocks = (per_split / 4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
  wgrad_accumulate_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(parts, splits, per_split, cols, row_scale, grad);
  return check_launch("relnet_wgrad_accumulate");
}

// table: DEVICE array of n relnet_relayout_desc (tile_start = exclusive prefix of taps * tiles_co * tiles_ci), total_tiles = their sum
extern "C" int relnet_weight_relayout(const void* table, int n, int total_tiles, void* stream) {
  RELNET_REQUIRE(table && n > 0 && total_tiles > 0, "relnet_weight_relayout: empty table");
  weight_relayout_kernel<<<(unsigned)total_tiles, 256, 0, (hipStream_t)stream>>>((const relnet::RelayoutProblem*)table, n);
  return relnet::check_launch("relnet_weight_relayout");
}

// table: DEVICE array of n relnet_fragpack_desc, total_blocks = sum of ceil((N / 32) (K / 16) 64 / 256) over the table
extern "C" int relnet_weight_fragpack(const void* table, int n, int total_blocks, void* stream) {
  RELNET_REQUIRE(table && n > 0 && total_blocks > 0, "relnet_weight_fragpack: empty table");
  weight_fragpack_kernel<<<(unsigned)total_blocks, 256, 0, (hipStream_t)stream>>>((const relnet::FragPackProblem*)table, n);
  return relnet::check_launch("relnet_weight_fragpack");
}

extern "C" int relnet_relation_bwd_pack(const float* dq, const float* dk, const float* dvw, void* out, int B, int N, int M, int d,
                                        void* stream) {
  RELNET_REQUIRE(dq && dk && dvw && out && B > 0 && N > 0 && M > 0 && M <= N && d > 0 && d % 8 == 0,
                 "relnet_relation_bwd_pack: bad arguments (B=%d N=%d M=%d d=%d)", B, N, M, d);
  RELNET_REQUIRE((((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dvw | (uintptr_t)out) & 15) == 0, "relnet_relation_bwd_pack: operands must be 16-byte aligned");
  const long total8 = (long)B * N * (3 * d / 8);
  long blocks = (total8 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  relation_bwd_pack_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(dq, dk, dvw, (unsigned short*)out, N, M, d, total8);
  return relnet::check_launch("relnet_relation_bwd_pack");
}

extern "C" int relnet_lnms_scatter_bwd(const float* d_sorted, const int* rank_idx, float* d_prob, int B, int N, int C, int F, void* stream) {
  RELNET_REQUIRE(d_sorted && rank_idx && d_prob && B > 0 && N > 0 && C > 0 && F > 0, "relnet_lnms_scatter_bwd: bad arguments");
  const long total = (long)B * C * F;
  lnms_scatter_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(d_sorted, rank_idx, d_prob, B, N, C, F);
  return relnet::check_launch("relnet_lnms_scatter_bwd");
}
