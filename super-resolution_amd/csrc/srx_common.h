// Internal helpers shared by the libsrx translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "srx.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

void srx_set_error(const char* fmt, ...);

#define SRX_CHECK_ARG(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      srx_set_error(__VA_ARGS__);           \
      return SRX_ERR_INVALID;               \
    }                                       \
  } while (0)

#define SRX_CHECK_HIP(expr)                                                        \
  do {                                                                             \
    hipError_t e_ = (expr);                                                        \
    if (e_ != hipSuccess) {                                                        \
      srx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SRX_ERR_HIP;                                                          \
    }                                                                              \
  } while (0)

// A kernel attribute (the dynamic LDS limit) belongs to ONE device's copy of the kernel:
// hipFuncSetAttribute applies to the device that is current when it is called.  One flag per
// device, one array per kernel instantiation (a function-local static of its launcher).
#define SRX_MAX_DEVICES 64
struct srx_per_device_once {
  unsigned char done[SRX_MAX_DEVICES];    // zero-initialised (static storage)
};
// Runs set(arg) the first time it is called for `device` with this flag set; SRX_OK afterwards.
// Two threads racing on a device's first launch both run set(): it is idempotent.
static inline int srx_once_per_device(srx_per_device_once* flags, int device, int (*set)(void*), void* arg) {
  SRX_CHECK_ARG(device >= 0 && device < SRX_MAX_DEVICES, "device index %d out of range [0, %d)", device, SRX_MAX_DEVICES);
  if (__atomic_load_n(&flags->done[device], __ATOMIC_ACQUIRE)) return SRX_OK;
  const int rc = set(arg);
  if (rc == SRX_OK) __atomic_store_n(&flags->done[device], (unsigned char)1, __ATOMIC_RELEASE);
  return rc;
}
struct srx_lds_attr { const void* kern; int bytes; };
static inline int srx_set_lds_attr(void* a) {
  const srx_lds_attr* s = static_cast<const srx_lds_attr*>(a);
  SRX_CHECK_HIP(hipFuncSetAttribute(s->kern, hipFuncAttributeMaxDynamicSharedMemorySize, s->bytes));
  return SRX_OK;
}
// The dynamic LDS limit of `kern` on the current device, set once per device.
static inline int srx_ensure_dynamic_lds(srx_per_device_once* flags, const void* kern, int bytes) {
  int dev = 0;
  SRX_CHECK_HIP(hipGetDevice(&dev));
  srx_lds_attr a = {kern, bytes};
  return srx_once_per_device(flags, dev, srx_set_lds_attr, &a);
}

static inline int srx_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    srx_set_error("launch of %s failed: %s", what, hipGetErrorString(e));
    return SRX_ERR_HIP;
  }
  return SRX_OK;
}

// ---- element traits -------------------------------------------------------
template <typename T> struct Oct;   // 8 elements of T, 16-byte pieces
template <> struct Oct<bf16> {
  static constexpr int PPO = 1;     // 16-byte pieces per octet
  typedef bf16x8 vec;
};
template <> struct Oct<float> {
  static constexpr int PPO = 2;
  typedef f32x8 vec;
};

__device__ __forceinline__ void load_oct(const bf16* p, float v[8]) {
  bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void load_oct(const float* p, float v[8]) {
  f32x4 a = *reinterpret_cast<const f32x4*>(p);
  f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
__device__ __forceinline__ void store_oct(bf16* p, const float v[8]) {
  bf16x8 t;
#pragma unroll
  for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
  *reinterpret_cast<bf16x8*>(p) = t;
}
__device__ __forceinline__ void store_oct(float* p, const float v[8]) {
  f32x4 a, b;
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
  *reinterpret_cast<f32x4*>(p) = a;
  *reinterpret_cast<f32x4*>(p + 4) = b;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Row r (0..31) of a 32-row MFMA output block -> position of the channel inside
// the block.  The C/D map of v_mfma_f32_32x32x*: lane half h, register i holds
// row (i&3) + 8*(i>>2) + 4*h.  Rows are assigned so that registers 0..7 and
// 8..15 of a lane each form one stored octet.
//  perm 0 (plain):  register i of half h -> channel 8*(2*(i>>3)+h) + (i&7)
//  perm 1 (PixelShuffle(2)): register i of half h -> conv channel 4*(i&7) + 2*h + (i>>3)
__host__ __device__ __forceinline__ int srx_row_to_chan(int row, int perm) {
  int h = (row >> 2) & 1;
  int i = (row & 3) | ((row >> 3) << 2);
  return perm == 0 ? 8 * (2 * (i >> 3) + h) + (i & 7) : 4 * (i & 7) + 2 * h + (i >> 3);
}
