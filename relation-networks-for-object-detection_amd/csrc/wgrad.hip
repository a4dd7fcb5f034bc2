// Weight gradients of the training step:  dW[Cout][K] += s_row^2 * sum_p dY[p][Cout] * X[p][K]   (p = output pixels / rois)
//
// MXNet derives these products by autograd from the symbols (Convolution / FullyConnected backward); the operands are the
// tensors this repository keeps in HBM anyway: dY and X are bf16 "pixel-major" maps ([P][C], channels contiguous), i.e. BOTH
// have the contraction index as their slow dimension ("TN" GEMM).  Round 2 transposed both operands (1 490 transpose launches
// per step), wrote fp32 split-K partials and summed them in a third kernel.  This kernel does it in one pass:
//   * tile 128 (Cout) x 128 (K columns) per 4-wave workgroup, 64-pixel slabs of both operands staged in LDS exactly as they
//     lie in memory ([pixel][channel], 16-byte global loads, ds_write_b128);
//   * MFMA fragments come out of LDS TRANSPOSED by gfx950's ds_read_b64_tr_b16 (a 16-lane group reads a [4 pixel][16 channel]
//     block and every lane receives the 4 pixels of ITS channel): no register shuffles, no transposed copies;
//   * 3x3 (dilated) and strided 1x1 convolutions gather their X rows on the fly (implicit im2col: column c of the K axis is
//     tap c / Cin, channel c % Cin; out-of-image taps read zero), so the [P][9 Cin] patch matrix is never written;
//   * split-K over the pixels fills the chip (Cout x K is at most a few hundred tiles, P is 2 464 .. 75 000); every workgroup
//     adds its tile into the fp32 gradient buffer with hardware float atomics (global_atomic_add_f32), the folded BatchNorm
//     factor s^2 applied on the way.  Summation order is therefore not fixed: results differ in the last bits between runs.
// Row pitch of an LDS slab is 128 + 32 elements (320 B): the four pixel rows of one transposed read start 80 banks apart
// (16 banks mod 64), so the 2 x 32-lane halves of the instruction are conflict free.
#include "common.h"

namespace relnet {

typedef __bf16 v4bf16 __attribute__((__vector_size__(4 * sizeof(__bf16))));
#define RELNET_LDS __attribute__((address_space(3)))

constexpr int kWgBM = 128, kWgBN = 128, kWgBP = 64;      // tile rows (Cout), tile columns (K), pixels per slab
constexpr int kWgLd = 128 + 32;                          // LDS row pitch in elements

struct WgradArgs {
  const unsigned short* dy; long dy_ld; int dy_cols;     // [P][dy_cols <= dy_ld]  (columns >= Cout are zero padding)
  const unsigned short* x; long x_pix;                   // activation, element stride between pixels (channels contiguous)
  float* dw; long dw_ld;                                 // [Cout][Ktot] fp32 accumulator
  const float* row_scale;                                // optional [Cout]
  int P, Cout, Ktot, Cin;
  int conv, ks, stride, dil, pad;                        // conv = 0: X row of pixel p is row p
  int Hout, Wout, Hin, Win;
  int tiles_m, tiles_n, splits, chunk;                   // chunk = pixels per split (multiple of kWgBP)
  int debug_plain;                                       // 1: assemble the fragments with scalar LDS reads (test aid)
};

__device__ __forceinline__ uint4 ldg16(const unsigned short* p) { return *(const uint4*)p; }

// Fragment of one MFMA operand for k-step kk: lane (l31, half) <- pixels 16 kk + 8 half + 0..7 of channel ch0 + l31.
__device__ __forceinline__ bf16x8 frag_tr(const unsigned short* slab, int kk, int ch0, int lane) {
  const int g = lane >> 4, j = lane & 15, half = g >> 1;
  const int ch = ch0 + 16 * (g & 1) + 4 * (j & 3);
  const int px = 16 * kk + 8 * half + (j >> 2);
  const unsigned short* p0 = slab + px * kWgLd + ch;
  const v4bf16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((RELNET_LDS v4bf16*)(p0));
  const v4bf16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((RELNET_LDS v4bf16*)(p0 + 4 * kWgLd));
  union { struct { v4bf16 a, b; } v; bf16x8 f; } u;
  u.v.a = lo; u.v.b = hi;
  return u.f;
}

__device__ __forceinline__ bf16x8 frag_plain(const unsigned short* slab, int kk, int ch0, int lane) {
  const int l31 = lane & 31, half = lane >> 5;
  bf16x8 f;
#pragma unroll
  for (int t = 0; t < 8; ++t) f[t] = (short)slab[(16 * kk + 8 * half + t) * kWgLd + ch0 + l31];
  return f;
}

__global__ __launch_bounds__(256, 2) void wgrad_tn_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned short sA[kWgBP * kWgLd];
  __shared__ __attribute__((aligned(16))) unsigned short sB[kWgBP * kWgLd];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  // all tiles of one pixel chunk on one XCD (blocks go to XCD id % 8): the chunk's dY / X rows are then fetched once per L2
  const int tiles = a.tiles_m * a.tiles_n;
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3;
  const int split = xcd + 8 * (jj / tiles), tile = jj % tiles;
  if (split >= a.splits) return;
  const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
  const int m0 = tm * kWgBM, n0 = tn * kWgBN;
  const int p_begin = split * a.chunk;
  const int p_end = min(a.P, p_begin + a.chunk);
  if (p_begin >= p_end) return;

  // this thread's share of a slab: column chunk cc (8 elements), rows r0 + 16 i
  const int cc = tid & 15, r0 = tid >> 4;
  const int a_col = m0 + cc * 8;
  const bool a_ok = a_col < a.dy_cols;
  const int b_col = n0 + cc * 8;
  const bool b_ok = b_col < a.Ktot;
  int tap_r = 0, tap_s = 0, cin = b_col;
  if (a.conv) {
    const int tap = b_col / a.Cin;
    cin = b_col - tap * a.Cin;
    tap_r = tap / a.ks; tap_s = tap - tap_r * a.ks;
  }
  const int hw = a.Hout * a.Wout;
  const int off_y = tap_r * a.dil - a.pad, off_x = tap_s * a.dil - a.pad;

  uint4 ra[4], rb[4];
  auto fetch = [&](int p0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = p0 + r0 + 16 * i;
      const bool in = p < p_end;
      ra[i] = make_uint4(0u, 0u, 0u, 0u);
      rb[i] = make_uint4(0u, 0u, 0u, 0u);
      if (in && a_ok) ra[i] = ldg16(a.dy + (long)p * a.dy_ld + a_col);
      if (in && b_ok) {
        if (!a.conv) {
          rb[i] = ldg16(a.x + (long)p * a.x_pix + cin);
        } else {
          const int bimg = p / hw, rem = p - bimg * hw;
          const int y = rem / a.Wout, xx = rem - y * a.Wout;
          const int sy = y * a.stride + off_y, sx = xx * a.stride + off_x;
          if ((unsigned)sy < (unsigned)a.Hin && (unsigned)sx < (unsigned)a.Win)
            rb[i] = ldg16(a.x + ((long)(bimg * a.Hin + sy) * a.Win + sx) * a.x_pix + cin);
        }
      }
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  fetch(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += kWgBP) {
    if (p0 > p_begin) __syncthreads();                     // everybody is done reading the previous slab
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *(uint4*)(sA + (r0 + 16 * i) * kWgLd + cc * 8) = ra[i];
      *(uint4*)(sB + (r0 + 16 * i) * kWgLd + cc * 8) = rb[i];
    }
    __syncthreads();
    if (p0 + kWgBP < p_end) fetch(p0 + kWgBP);             // next slab's loads fly under this slab's MFMAs
#pragma unroll
    for (int kk = 0; kk < kWgBP / 16; ++kk) {
      bf16x8 fa[2], fb[2];
      if (a.debug_plain) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[mi] = frag_plain(sA, kk, 64 * wm + 32 * mi, lane);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[ni] = frag_plain(sB, kk, 64 * wn + 32 * ni, lane);
      } else {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) fa[mi] = frag_tr(sA, kk, 64 * wm + 32 * mi, lane);
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) fb[ni] = frag_tr(sB, kk, 64 * wn + 32 * ni, lane);
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
  }

  // epilogue: dW[m][n] += s_m^2 * acc   (hardware float atomics; rows = Cout, columns contiguous over the lanes)
  const int l31 = lane & 31, half = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
      if (row >= a.Cout) continue;
      float s2 = 1.f;
      if (a.row_scale) { const float s = a.row_scale[row]; s2 = s * s; }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int col = n0 + 64 * wn + 32 * ni + l31;
        if (col < a.Ktot) unsafeAtomicAdd(a.dw + (long)row * a.dw_ld + col, s2 * acc[mi][ni][r]);
      }
    }
}

// probe of the transposed LDS read (test aid): out[lane][mode][0..3] for lane-linear addresses over value == index
__global__ __launch_bounds__(64) void tr_probe_kernel(unsigned short* out) {
  __shared__ __attribute__((aligned(16))) unsigned short s[1024];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) s[i] = (unsigned short)i;
  __syncthreads();
  const v4bf16 v = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((RELNET_LDS v4bf16*)(s + lane * 4));
  union { v4bf16 v; unsigned short u[4]; } c;
  c.v = v;
#pragma unroll
  for (int t = 0; t < 4; ++t) out[lane * 4 + t] = c.u[t];
}

}  // namespace relnet

using namespace relnet;

static int g_wgrad_debug_plain = 0;
extern "C" void relnet_wgrad_debug_plain(int on) { g_wgrad_debug_plain = on; }

extern "C" int relnet_debug_tr_probe(unsigned short* out256, void* stream) {
  RELNET_REQUIRE(out256, "relnet_debug_tr_probe: null output");
  tr_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out256);
  return check_launch("relnet_debug_tr_probe");
}

// dw [Cout][Ktot] fp32 (row pitch dw_ld) += row_scale^2 * dY^T X.
//   dy  [P][dy_cols] bf16 (row pitch dy_ld; columns >= Cout must be zero or absent), P = B * Hout * Wout pixels (or rows)
//   x   bf16 activation with element stride x_pix between pixels; plain product (ks = 1, stride = 1): row p of X is pixel p
//       and Ktot = Cin; convolution: [B][Hin][Win] pixels, Ktot = ks * ks * Cin in (tap, channel) order = pack_conv_weight order
extern "C" int relnet_wgrad(const void* dy, long dy_ld, int dy_cols, const void* x, long x_pix, float* dw, long dw_ld,
                            const float* row_scale, int P, int Cout, int Cin, int ks, int stride, int dil, int pad,
                            int B, int Hout, int Wout, int Hin, int Win, void* stream) {
  RELNET_REQUIRE(dy && x && dw, "relnet_wgrad: null operand");
  RELNET_REQUIRE(P > 0 && Cout > 0 && Cin > 0 && ks >= 1 && stride >= 1, "relnet_wgrad: bad shape");
  RELNET_REQUIRE(dy_ld % 8 == 0 && dy_cols % 8 == 0 && x_pix % 8 == 0 && Cin % 8 == 0,
                 "relnet_wgrad: rows must be 16-byte aligned (dy_ld %ld, dy_cols %d, x_pix %ld, Cin %d)", dy_ld, dy_cols, x_pix, Cin);
  RELNET_REQUIRE(((uintptr_t)dy & 15) == 0 && ((uintptr_t)x & 15) == 0, "relnet_wgrad: operands must be 16-byte aligned");
  const bool conv = !(ks == 1 && stride == 1);
  if (conv) RELNET_REQUIRE(B > 0 && (long)B * Hout * Wout == P && Hin > 0 && Win > 0, "relnet_wgrad: P %d != B * Hout * Wout", P);
  WgradArgs a;
  a.dy = (const unsigned short*)dy; a.dy_ld = dy_ld; a.dy_cols = dy_cols; a.x = (const unsigned short*)x; a.x_pix = x_pix;
  a.dw = dw; a.dw_ld = dw_ld; a.row_scale = row_scale; a.P = P; a.Cout = Cout; a.Ktot = ks * ks * Cin; a.Cin = Cin;
  a.conv = conv ? 1 : 0; a.ks = ks; a.stride = stride; a.dil = dil; a.pad = pad;
  a.Hout = conv ? Hout : 1; a.Wout = conv ? Wout : P; a.Hin = Hin; a.Win = Win;
  a.tiles_m = (Cout + kWgBM - 1) / kWgBM; a.tiles_n = (a.Ktot + kWgBN - 1) / kWgBN;
  const int tiles = a.tiles_m * a.tiles_n;
  // enough workgroups for 256 CUs x 2-3 resident, at least 4 slabs each; splits a multiple of 8 (one XCD per pixel chunk)
  int splits = (768 + tiles - 1) / tiles;
  const int max_splits = (P + 4 * kWgBP - 1) / (4 * kWgBP);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int chunk = (P + splits - 1) / splits;
  chunk = (chunk + kWgBP - 1) / kWgBP * kWgBP;
  splits = (P + chunk - 1) / chunk;
  a.splits = splits; a.chunk = chunk; a.debug_plain = g_wgrad_debug_plain;
  const int groups = (splits + 7) / 8;
  wgrad_tn_kernel<<<(unsigned)(8 * groups * tiles), 256, 0, (hipStream_t)stream>>>(a);
  return check_launch("relnet_wgrad");
}
