// Weight gradients of the training step:  dW[Cout][K] += s_row^2 * sum_p dY[p][Cout] * X[p][K]   (p = output pixels / rois)
//
// MXNet derives these products by autograd from the symbols (Convolution / FullyConnected backward); the operands are the
// tensors this repository keeps in HBM anyway: dY and X are bf16 "pixel-major" maps ([P][C], channels contiguous), i.e. BOTH
// have the contraction index as their slow dimension ("TN" GEMM).  Round 2 transposed both operands (1 490 transpose launches
// per step), wrote fp32 split-K partials and summed them in a third kernel.  Here:
//   * MFMA fragments come out of LDS TRANSPOSED by gfx950's ds_read_b64_tr_b16 (a 16-lane group reads a [4 pixel][16 channel]
//     block and every lane receives the 4 pixels of ITS channel): the slabs are staged exactly as they lie in memory
//     ([pixel][channel], 16-byte global loads, ds_write_b128), no register shuffles, no transposed copies;
//   * 3x3 (dilated) and strided 1x1 convolutions gather their X rows on the fly (implicit im2col: column c of the K axis is
//     tap c / Cin, channel c % Cin; out-of-image taps read zero), so the [P][9 Cin] patch matrix is never written;
//   * GROUPED + STREAM-K: one launch takes a table of layers (e.g. the 69 convolutions of res4, whose dY / X all exist once
//     the stage's data-gradient chain has run).  The work unit is (layer, 256 x 256 output tile, 64-pixel slab); the units
//     of the whole table are dealt to one persistent 8-wave workgroup per CU in equal contiguous shares, so every CU is
//     busy whatever the layer shapes, and a tile is combined across workgroups only where a share boundary cuts it:
//     ~1.7 atomic tile flushes per tile instead of one per (tile, pixel split) -- measured with per-layer launches:
//     atomics 3.1 ms of a 9 ms total at 8 images, and a 128 x 128 tile bound by LDS WRITE bandwidth (every slab element
//     written once, read by two waves only);
//   * wave tile 64 x 128 (2 + 4 fragments per 8 MFMAs), two LDS slab buffers: slab s+1 is written and slab s+2's global
//     loads are issued before slab s is consumed; one barrier per slab.
// The folded BatchNorm factor s^2 is applied in the flush; summation order across share boundaries is not fixed.
// Row pitch of an LDS slab is 256 + 32 elements (576 B = 144 banks = 16 mod 64): the four pixel rows of one transposed read
// start 16 banks apart, so the 2 x 32-lane halves of the instruction are conflict free.
#include "common.h"

namespace relnet {

typedef __bf16 v4bf16 __attribute__((__vector_size__(4 * sizeof(__bf16))));
#define RELNET_LDS __attribute__((address_space(3)))

constexpr int kWgBN = 256, kWgBP = 64;                   // tile columns (K), pixels per slab; tile rows (Cout) = 64 WM
constexpr int kWgLdB = kWgBN + 32;                       // LDS row pitch of the X slab in elements

struct WgradProblem {
  const unsigned short* dy; const unsigned short* x; float* dw; const float* row_scale;
  long dy_ld, x_pix, dw_ld;
  int dy_cols, P, Cout, Ktot, Cin;
  int conv, ks, stride, dil, pad;                        // conv = 0: X row of pixel p is row p
  int Hout, Wout, Hin, Win;
  int tiles_m, tiles_n, slabs;                           // output tiles and 64-pixel slabs of this problem
  int unit_start;                                        // first work unit of this problem in the launch
  int tile_start;                                        // first output tile of this problem in the launch (tile-aligned shares)
};

struct WgradTableChunk { WgradProblem p[16]; int n, offset; };

__global__ void wgrad_fill_table_kernel(WgradTableChunk c, WgradProblem* table) {
  const int i = threadIdx.x;
  if (i < c.n) table[c.offset + i] = c.p[i];
}

__device__ __forceinline__ uint4 ldg16(const unsigned short* p) { return *(const uint4*)p; }

// Fragment of one MFMA operand for k-step kk: lane (l31, half) <- pixels 16 kk + 8 half + 0..7 of channel ch0 + l31.
template <int LD>
__device__ __forceinline__ bf16x8 frag_tr(const unsigned short* slab, int kk, int ch0, int lane) {
  const int g = lane >> 4, j = lane & 15, half = g >> 1;
  const int ch = ch0 + 16 * (g & 1) + 4 * (j & 3);
  const int px = 16 * kk + 8 * half + (j >> 2);
  const unsigned short* p0 = slab + px * LD + ch;
  const v4bf16 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((RELNET_LDS v4bf16*)(p0));
  const v4bf16 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((RELNET_LDS v4bf16*)(p0 + 4 * LD));
  union { struct { v4bf16 a, b; } v; bf16x8 f; } u;
  u.v.a = lo; u.v.b = hi;
  return u.f;
}

template <int LD>
__device__ __forceinline__ bf16x8 frag_plain(const unsigned short* slab, int kk, int ch0, int lane) {
  const int l31 = lane & 31, half = lane >> 5;
  bf16x8 f;
#pragma unroll
  for (int t = 0; t < 8; ++t) f[t] = (short)slab[(16 * kk + 8 * half + t) * LD + ch0 + l31];
  return f;
}

// WM = wavefront rows: tile = 64 WM (Cout) x 256 (K columns), 2 WM wavefronts of 64 x 128.
// mode (measurement aids): 1 no flush, 4 no global loads; plain: fragments by scalar LDS reads (test aid).
template <int WM, bool PLAIN>
__global__ __launch_bounds__(128 * WM) void wgrad_streamk_kernel(const WgradProblem* __restrict__ table, int nprob, int total_units,
                                                                 int mode, int total_tiles) {
  constexpr int kBM = 64 * WM, kLdA = kBM + 32, NT = 128 * WM;
  constexpr int kAchunks = kBM / 8, kArows = NT / kAchunks, kApass = kWgBP / kArows;     // dY slab: rows per pass, passes
  constexpr int kBchunks = kWgBN / 8, kBrows = NT / kBchunks, kBpass = kWgBP / kBrows;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* sA0 = smem;                                   // [2][64][kLdA]
  unsigned short* sB0 = smem + 2 * kWgBP * kLdA;                // [2][64][kWgLdB]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WM, wn = wave / WM;
  const int l31 = lane & 31, half = lane >> 5;
  // this workgroup's share of the work units
  // total_tiles > 0 (round 6, opt-in): the shares are cut at TILE boundaries -- a workgroup owns whole output tiles, so the flush below is a plain
  // read-modify-write instead of float atomics (deterministic; slower than the balanced shares: see relnet_wgrad_grouped).
  const long G = gridDim.x;
  int u, u_end;
  const bool whole_tiles = total_tiles > 0;
  if (whole_tiles) {
    auto tile_unit = [&](int t) {
      if (t >= total_tiles) return total_units;
      int lo = 0, hi = nprob - 1;
      while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid].tile_start <= t) lo = mid; else hi = mid - 1; }
      return table[lo].unit_start + (t - table[lo].tile_start) * table[lo].slabs;
    };
    u = tile_unit((int)(((long)total_tiles * blockIdx.x) / G));
    u_end = tile_unit((int)(((long)total_tiles * (blockIdx.x + 1)) / G));
  } else {
    u = (int)(((long)total_units * blockIdx.x) / G);
    u_end = (int)(((long)total_units * (blockIdx.x + 1)) / G);
  }
  const int ca = tid % kAchunks, ra0 = tid / kAchunks;
  const int cb = tid % kBchunks, rb0 = tid / kBchunks;
  const bool no_loads = (mode & 4) != 0;

  while (u < u_end) {
    // ---- locate (problem, tile, first slab) of unit u
    int lo = 0, hi = nprob - 1;
    while (lo < hi) {                                           // last problem with unit_start <= u
      const int mid = (lo + hi + 1) >> 1;
      if (table[mid].unit_start <= u) lo = mid; else hi = mid - 1;
    }
    const WgradProblem a = table[lo];
    const int rel = u - a.unit_start;
    const int tile = rel / a.slabs, s0 = rel - tile * a.slabs;
    const int s1 = min(a.slabs, s0 + (u_end - u));              // slabs [s0, s1) of this tile are ours
    u += s1 - s0;
    const int tm = tile % a.tiles_m, tn = tile / a.tiles_m;
    const int m0 = tm * kBM, n0 = tn * kWgBN;
    const int p_begin = s0 * kWgBP, p_end = min(a.P, s1 * kWgBP);

    const int a_col = m0 + ca * 8;
    const bool a_ok = a_col < a.dy_cols;
    const int b_col = n0 + cb * 8;
    const bool b_ok = b_col < a.Ktot;
    int tap_r = 0, tap_s = 0, cin = b_col;
    if (a.conv) {
      const int tap = b_col / a.Cin;
      cin = b_col - tap * a.Cin;
      tap_r = tap / a.ks; tap_s = tap - tap_r * a.ks;
    }
    const int hw = a.Hout * a.Wout;
    const int off_y = tap_r * a.dil - a.pad, off_x = tap_s * a.dil - a.pad;

    uint4 ra[kApass], rb[kBpass];
    // implicit im2col rows: (image, y, x) of this thread's pixel of every pass, divided out ONCE per share and then advanced by
    // 64 pixels per slab with adds / compares (two integer divisions per 16-byte load made the address stream of the 3x3
    // problems cost as many VALU cycles as the slab's MFMAs)
    int cimg[kBpass], cyx[kBpass];                           // y << 16 | x
    const int step_y = kWgBP / a.Wout, step_x = kWgBP - step_y * a.Wout;
    if (a.conv) {
#pragma unroll
      for (int i = 0; i < kBpass; ++i) {
        const int p = p_begin + rb0 + kBrows * i;
        const int bimg = p / hw, rem = p - bimg * hw;
        const int y = rem / a.Wout;
        cimg[i] = bimg; cyx[i] = (y << 16) | (rem - y * a.Wout);
      }
    }
    auto advance = [&]() {
#pragma unroll
      for (int i = 0; i < kBpass; ++i) {
        int y = (cyx[i] >> 16) + step_y, xx = (cyx[i] & 0xffff) + step_x;
        if (xx >= a.Wout) { xx -= a.Wout; ++y; }
        while (y >= a.Hout) { y -= a.Hout; ++cimg[i]; }
        cyx[i] = (y << 16) | xx;
      }
    };
    auto fetch = [&](int p0) {
#pragma unroll
      for (int i = 0; i < kApass; ++i) {
        const int p = p0 + ra0 + kArows * i;
        ra[i] = make_uint4(0u, 0u, 0u, 0u);
        if (p < p_end && a_ok && !no_loads) ra[i] = ldg16(a.dy + (long)p * a.dy_ld + a_col);
      }
#pragma unroll
      for (int i = 0; i < kBpass; ++i) {
        const int p = p0 + rb0 + kBrows * i;
        rb[i] = make_uint4(0u, 0u, 0u, 0u);
        if (p < p_end && b_ok && !no_loads) {
          if (!a.conv) {
            rb[i] = ldg16(a.x + (long)p * a.x_pix + cin);
          } else {
            const int bimg = cimg[i], y = cyx[i] >> 16, xx = cyx[i] & 0xffff;
            const int sy = y * a.stride + off_y, sx = xx * a.stride + off_x;
            if ((unsigned)sy < (unsigned)a.Hin && (unsigned)sx < (unsigned)a.Win)
              rb[i] = ldg16(a.x + ((long)(bimg * a.Hin + sy) * a.Win + sx) * a.x_pix + cin);
          }
        }
      }
    };
    auto stage = [&](int buf) {
      unsigned short* sA = sA0 + buf * kWgBP * kLdA;
      unsigned short* sB = sB0 + buf * kWgBP * kWgLdB;
#pragma unroll
      for (int i = 0; i < kApass; ++i) *(uint4*)(sA + (ra0 + kArows * i) * kLdA + ca * 8) = ra[i];
#pragma unroll
      for (int i = 0; i < kBpass; ++i) *(uint4*)(sB + (rb0 + kBrows * i) * kWgLdB + cb * 8) = rb[i];
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // prologue: slab 0 staged, slab 1 in flight.  (The barrier also protects the buffers against the previous tile's readers.)
    fetch(p_begin);
    if (a.conv) advance();
    __syncthreads();
    stage(0);
    if (p_begin + kWgBP < p_end) { fetch(p_begin + kWgBP); if (a.conv) advance(); }
    __syncthreads();
    int buf = 0;
    for (int p0 = p_begin; p0 < p_end; p0 += kWgBP) {
      // slab s+1 (its loads were issued a whole slab ago) -> the other buffer; slab s+2's loads start now
      if (p0 + kWgBP < p_end) {
        stage(buf ^ 1);
        if (p0 + 2 * kWgBP < p_end) { fetch(p0 + 2 * kWgBP); if (a.conv) advance(); }
      }
      const unsigned short* sA = sA0 + buf * kWgBP * kLdA;
      const unsigned short* sB = sB0 + buf * kWgBP * kWgLdB;
#pragma unroll
      for (int kk = 0; kk < kWgBP / 16; ++kk) {
        bf16x8 fa[2], fb[4];
        if constexpr (PLAIN) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) fa[mi] = frag_plain<kLdA>(sA, kk, 64 * wm + 32 * mi, lane);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) fb[ni] = frag_plain<kWgLdB>(sB, kk, 128 * wn + 32 * ni, lane);
        } else {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) fa[mi] = frag_tr<kLdA>(sA, kk, 64 * wm + 32 * mi, lane);
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) fb[ni] = frag_tr<kWgLdB>(sB, kk, 128 * wn + 32 * ni, lane);
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
      buf ^= 1;
    }

    // ---- flush: dW[m][n] += s_m^2 * acc  (hardware float atomics: a tile may be shared with the neighbouring workgroups;
    //      columns are contiguous over the lanes)
    if (mode & 1) continue;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 64 * wm + 32 * mi + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row >= a.Cout) continue;
        float s2 = 1.f;
        if (a.row_scale) { const float s = a.row_scale[row]; s2 = s * s; }
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int col = n0 + 128 * wn + 32 * ni + l31;
          if (col < a.Ktot) {
            float* dst = a.dw + (long)row * a.dw_ld + col;
            if (whole_tiles) *dst += s2 * acc[mi][ni][r];            // this workgroup is the tile's only writer in this launch
            else unsafeAtomicAdd(dst, s2 * acc[mi][ni][r]);
          }
        }
      }
  }
}

// probe of the transposed LDS read (test aid): out[lane][0..3] for lane-linear addresses over value == index
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

static int g_wgrad_debug_plain = 0, g_wgrad_blocks = 0, g_wgrad_mode = 0, g_wgrad_wm = 0, g_wgrad_tiles = 0;
extern "C" void relnet_wgrad_debug_tiles(int mode) { g_wgrad_tiles = mode; }
extern "C" void relnet_wgrad_debug_plain(int on) { g_wgrad_debug_plain = on; }
// measurement knobs (tools/bench_wgrad.py): persistent workgroups per launch (0 = one per CU), ablation mode bits, forced
// wavefront rows (0 = by shape)
extern "C" void relnet_wgrad_tune(int workgroups, int mode, int wm) { g_wgrad_blocks = workgroups; g_wgrad_mode = mode; g_wgrad_wm = wm; }

extern "C" int relnet_debug_tr_probe(unsigned short* out256, void* stream) {
  RELNET_REQUIRE(out256, "relnet_debug_tr_probe: null output");
  tr_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out256);
  return check_launch("relnet_debug_tr_probe");
}

// One layer of a grouped launch, as the caller describes it (host struct of the C-ABI, see include/relnet_hip.h).
struct relnet_wgrad_desc {
  const void* dy; long dy_ld; int dy_cols;
  const void* x; long x_pix;
  float* dw; long dw_ld;
  const float* row_scale;
  int P, Cout, Cin, ks, stride, dil, pad, B, Hout, Wout, Hin, Win;
};

static int wgrad_fill(const relnet_wgrad_desc& d, int bm, WgradProblem* out) {
  RELNET_REQUIRE(d.dy && d.x && d.dw, "relnet_wgrad: null operand");
  RELNET_REQUIRE(d.P > 0 && d.Cout > 0 && d.Cin > 0 && d.ks >= 1 && d.stride >= 1, "relnet_wgrad: bad shape");
  RELNET_REQUIRE(d.dy_ld % 8 == 0 && d.dy_cols % 8 == 0 && d.x_pix % 8 == 0 && d.Cin % 8 == 0,
                 "relnet_wgrad: rows must be 16-byte aligned (dy_ld %ld, dy_cols %d, x_pix %ld, Cin %d)", d.dy_ld, d.dy_cols, d.x_pix, d.Cin);
  RELNET_REQUIRE(((uintptr_t)d.dy & 15) == 0 && ((uintptr_t)d.x & 15) == 0, "relnet_wgrad: operands must be 16-byte aligned");
  const bool conv = !(d.ks == 1 && d.stride == 1);
  if (conv) RELNET_REQUIRE(d.B > 0 && (long)d.B * d.Hout * d.Wout == d.P && d.Hin > 0 && d.Win > 0, "relnet_wgrad: P %d != B * Hout * Wout", d.P);
  WgradProblem& a = *out;
  a.dy = (const unsigned short*)d.dy; a.dy_ld = d.dy_ld; a.dy_cols = d.dy_cols; a.x = (const unsigned short*)d.x; a.x_pix = d.x_pix;
  a.dw = d.dw; a.dw_ld = d.dw_ld; a.row_scale = d.row_scale; a.P = d.P; a.Cout = d.Cout; a.Ktot = d.ks * d.ks * d.Cin; a.Cin = d.Cin;
  a.conv = conv ? 1 : 0; a.ks = d.ks; a.stride = d.stride; a.dil = d.dil; a.pad = d.pad;
  a.Hout = conv ? d.Hout : 1; a.Wout = conv ? d.Wout : d.P; a.Hin = d.Hin; a.Win = d.Win;
  a.tiles_m = (d.Cout + bm - 1) / bm; a.tiles_n = (a.Ktot + kWgBN - 1) / kWgBN;
  a.slabs = (d.P + kWgBP - 1) / kWgBP;
  return 0;
}

// dw_i [Cout][Ktot] fp32 (row pitch dw_ld) += row_scale^2 * dY_i^T X_i for n layers in ONE launch (stream-K over all of
// their (tile, slab) units).  descs: HOST array; table_workspace: device memory, >= relnet_wgrad_workspace_bytes(n) bytes,
// which must stay untouched until the launch has run (the table is filled by small kernels on the same stream: no host
// copy, capture safe).
extern "C" long relnet_wgrad_workspace_bytes(int n) { return (long)sizeof(WgradProblem) * (n > 0 ? n : 0); }

extern "C" int relnet_wgrad_grouped(const relnet_wgrad_desc* descs, int n, void* table_workspace, void* stream) {
  RELNET_REQUIRE(descs && n > 0 && table_workspace, "relnet_wgrad_grouped: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  // 128-row tiles only when no layer of the group has more rows (half of a 256-row tile would multiply zeros)
  int max_cout = 0;
  for (int i = 0; i < n; ++i) max_cout = descs[i].Cout > max_cout ? descs[i].Cout : max_cout;
  const int wm = g_wgrad_wm ? g_wgrad_wm : (max_cout > 128 ? 4 : 2);
  const int bm = 64 * wm;
  long units = 0, tiles = 0;
  int max_slabs = 0;
  WgradTableChunk c;
  c.n = 0; c.offset = 0;
  for (int i = 0; i < n; ++i) {
    WgradProblem& p = c.p[c.n];
    if (wgrad_fill(descs[i], bm, &p) != 0) return -1;
    RELNET_REQUIRE(units + (long)p.tiles_m * p.tiles_n * p.slabs < (1L << 31), "relnet_wgrad_grouped: too many work units");
    p.unit_start = (int)units;
    p.tile_start = (int)tiles;
    units += (long)p.tiles_m * p.tiles_n * p.slabs;
    tiles += (long)p.tiles_m * p.tiles_n;
    max_slabs = p.slabs > max_slabs ? p.slabs : max_slabs;
    if (++c.n == 16 || i == n - 1) {
      wgrad_fill_table_kernel<<<1, 64, 0, s>>>(c, (WgradProblem*)table_workspace);
      c.offset += c.n; c.n = 0;
    }
  }
  int blocks = g_wgrad_blocks > 0 ? g_wgrad_blocks : 256;
  if (units < blocks) blocks = (int)units;
  // whole-tile shares (no atomics, bit-identical reruns): OPT-IN (g_wgrad_tiles = 2).  Built in round 6 on the estimate that the atomic flush was 0.5 of the
  // 0.77 ms the seven launches of a one-image step take; measured, same box (tools/scripts/r06_ab3.sh): one image 7.16 ms with stream-K shares, 8.14 ms with
  // whole tiles, two images 9.44 / 11.34 ms -- the fire-and-forget atomics overlap the next unit's loads, and what whole tiles add is up to one tile of
  // imbalance per workgroup (400 tiles on 256 workgroups = two rounds) plus the read of the read-modify-write.  Needs the layers of the group to accumulate
  // into disjoint memory (checked here; otherwise the atomics stay).  g_wgrad_tiles: 0 / 1 = stream-K shares, 2 = whole tiles whenever the writers are disjoint
  bool disjoint = true;
  for (int i = 0; i < n && disjoint; ++i)
    for (int j = i + 1; j < n && disjoint; ++j) {
      const float* a0 = descs[i].dw; const float* a1 = a0 + (long)(descs[i].Cout - 1) * descs[i].dw_ld + (long)descs[i].ks * descs[i].ks * descs[i].Cin;
      const float* b0 = descs[j].dw; const float* b1 = b0 + (long)(descs[j].Cout - 1) * descs[j].dw_ld + (long)descs[j].ks * descs[j].ks * descs[j].Cin;
      if (a0 < b1 && b0 < a1) disjoint = false;
    }
  const bool whole = disjoint && g_wgrad_tiles == 2;
  (void)max_slabs;
  if (whole && tiles < blocks) blocks = (int)tiles;
  const int total_tiles = whole ? (int)tiles : 0;
  const size_t lds = (size_t)2 * kWgBP * ((bm + 32) + kWgLdB) * 2;
  static relnet::PerDeviceOnce attr_once;
  if (attr_once.first()) {
    hipFuncSetAttribute((const void*)wgrad_streamk_kernel<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)wgrad_streamk_kernel<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)wgrad_streamk_kernel<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute((const void*)wgrad_streamk_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  }
  const WgradProblem* tb = (const WgradProblem*)table_workspace;
  if (g_wgrad_debug_plain) {
    if (wm == 4) wgrad_streamk_kernel<4, true><<<blocks, 512, lds, s>>>(tb, n, (int)units, g_wgrad_mode, total_tiles);
    else wgrad_streamk_kernel<2, true><<<blocks, 256, lds, s>>>(tb, n, (int)units, g_wgrad_mode, total_tiles);
  } else {
    if (wm == 4) wgrad_streamk_kernel<4, false><<<blocks, 512, lds, s>>>(tb, n, (int)units, g_wgrad_mode, total_tiles);
    else wgrad_streamk_kernel<2, false><<<blocks, 256, lds, s>>>(tb, n, (int)units, g_wgrad_mode, total_tiles);
  }
  return check_launch("relnet_wgrad_grouped");
}
