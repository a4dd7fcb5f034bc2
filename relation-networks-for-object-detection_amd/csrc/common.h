// Shared device helpers for the gfx950 (MI355X / CDNA4) kernels of the relation-network
// detection hot path.  gfx950 only: 64-wide wavefronts, MFMA 32x32x16 bf16 / 32x32x2 f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace relnet {

typedef __attribute__((ext_vector_type(8))) short bf16x8;    // 8 bf16 = MFMA A/B fragment
typedef __attribute__((ext_vector_type(4))) short bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;   // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int kWave = 64;

// float -> bf16 bits, round to nearest even (inputs are finite on this path).
__device__ __forceinline__ unsigned short f2bf(float f) {
  unsigned int u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) {
  return __uint_as_float(((unsigned int)h) << 16);
}
// two floats -> packed bf16x2 (round to nearest even): one v_cvt_pk_bf16_f32 on gfx950
typedef __bf16 hw_bf16x2 __attribute__((ext_vector_type(2)));
typedef float hw_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf16x2(float lo, float hi) {
  const hw_f32x2 v = {lo, hi};
  const hw_bf16x2 b = __builtin_convertvector(v, hw_bf16x2);
  return *(const unsigned int*)&b;
}

// Row index inside a 32x32 MFMA C/D tile held by this lane in accumulator register r:
//   col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)      (guide section 3)
__device__ __forceinline__ int mfma32_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

}  // namespace relnet

// Error plumbing of the C-ABI: every entry point returns 0 or a negative code and
// records a message retrievable with relnet_last_error().
extern "C" const char* relnet_last_error(void);
namespace relnet {
void set_error(const char* fmt, ...);
int check_launch(const char* what);
long device_cu_count();
}  // namespace relnet

namespace relnet {
// True exactly once per (call site, device): the "opt in to > 64 KiB dynamic LDS" attribute of a kernel is per DEVICE, so a
// process that drives several GPUs has to set it on each of them (thread-safe: the flag word is atomic; setting the
// attribute twice from two racing threads is harmless).
struct PerDeviceOnce {
  unsigned long long done = 0;
  bool first() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    return (__atomic_fetch_or(&done, bit, __ATOMIC_ACQ_REL) & bit) == 0;
  }
};
}  // namespace relnet

#define RELNET_REQUIRE(cond, ...)            \
  do {                                       \
    if (!(cond)) {                           \
      relnet::set_error(__VA_ARGS__);        \
      return -1;                             \
    }                                        \
  } while (0)
