// Shared helpers for the libldmk kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/ldmk.h"

namespace ldmk {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: %s", what, hipGetErrorString(e));
    return LDMK_EHIP;
  }
  return LDMK_OK;
}

// a stale error left in the thread by an earlier, unrelated HIP call must not be blamed on this launch
#define LDMK_ENTER() (void)hipGetLastError()

#define LDMK_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      ldmk::set_error(__VA_ARGS__);    \
      return LDMK_EINVAL;              \
    }                                  \
  } while (0)

#define LDMK_REQUIRE_MEM(cond, ...)    \
  do {                                 \
    if (!(cond)) {                     \
      ldmk::set_error(__VA_ARGS__);    \
      return LDMK_ENOMEM;              \
    }                                  \
  } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// x * sigmoid(x); v_exp + v_rcp (1 ulp each) instead of an IEEE division sequence
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// 64-lane butterfly reductions (wave = 64 on CDNA)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Bijective XCD-aware remap of a linear workgroup id: consecutive remapped ids land on the same
// XCD (blocks are dealt round-robin over 8 XCDs), so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int nx = 8;
  int q = nwg / nx, r = nwg % nx;
  int xcd = bid % nx, i = bid / nx;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + i;
}

}  // namespace ldmk
