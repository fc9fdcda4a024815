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

// GELU(g) = 0.5 g (1 + erf(g / sqrt 2)), the exact (erf) form of F.gelu, in 15 straight-line VALU instructions.
// erfc(z) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-z^2), t = 1 / (1 + p z), z >= 0 (Abramowitz & Stegun 7.1.26,
// |error| < 1.5e-7), and 1 + erf(x) = erfc(-x): with z = |g| / sqrt 2 and h = 0.5 g erfc(z) the result is g - h for
// g >= 0 and h for g < 0, i.e. max(g, 0) - |h| -- no cancellation in the negative tail and no branch.  Measured against
// float64 over [-12, 12]: max abs error 3.3e-7 (torch's own fp32 F.gelu: 1.2e-6).  ocml's erff is two divergent
// polynomial branches: the GEGLU epilogue of a 32x128 wave tile was 2927 VALU instructions for 32 outputs per lane.
__device__ __forceinline__ float gelu_erf_f(float g) {
  const float z = fabsf(g) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float p = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
  p = fmaf(p, t, 0.5f * 1.421413741f);
  p = fmaf(p, t, 0.5f * -0.284496736f);
  p = fmaf(p, t, 0.5f * 0.254829592f);
  p *= t;
  const float e = __builtin_amdgcn_exp2f(g * g * -0.72134752044448170368f);      // exp(-z^2) = 2^(-(g^2 / 2) log2 e)
  const float h = g * (p * e);                                                    // 0.5 g erfc(z), carries the sign of g
  return fmaxf(g, 0.f) - fabsf(h);
}

// LDMK_COMPUTE_F16X2: an operand outside the scaled fp16 range (|x| >= LDMK_F16X2_RANGE) raises the launch's range flag AND is
// saturated just inside the range before it is split.  The flagged launch's result is wrong either way and the caller computes it
// again in another arithmetic -- but it stays FINITE, so the launches behind it see plausible magnitudes and raise their own flags only
// for their own operands: one pass names every launch that has to change arithmetic, not everything downstream of the first one
// (an fp16 infinity would turn the rest of the network into NaNs, which trip every range check).  One v_med3_f32 per element.
#ifdef LDMK_NO_H2_CLAMP      /* A/B builds only (what the saturation costs): profiles/r05_ab_clamp.txt */
__device__ __forceinline__ float h2_clamp(float x) { return x; }
#else
__device__ __forceinline__ float h2_clamp(float x) { return __builtin_amdgcn_fmed3f(x, -999.99994f, 999.99994f); }
#endif
__device__ __forceinline__ float4 h2_clamp4(const float4& v) { return make_float4(h2_clamp(v.x), h2_clamp(v.y), h2_clamp(v.z), h2_clamp(v.w)); }

// LDMK_COMPUTE_F16X2: the LOW fp16 image of a pair, fp16(x' - hi), hi = the packed fp16 images of (x, y).  x' - hi is exact in fp32
// (hi is x' rounded to 11 bits), so v_fma_mix -- which reads the fp16 halves of hi in place and writes an fp16 half -- rounds once,
// exactly as converting the fp32 difference does: 2 instructions per pair instead of 2 widening converts + subtract(s) + a packed
// convert, and none of them a packed fp32 instruction (those wait for the wave's own MFMAs to drain and run at 40 % next to
// another wave's: tools/probe/valu_rate.hip, profiles/r05_valu_rate.txt).  -DLDMK_H2_SPLIT_CVT: the convert form (A/B)
__device__ __forceinline__ unsigned h2_lo_pair(unsigned h, float x, float y) {
#ifdef LDMK_H2_SPLIT_CVT
  typedef _Float16 h2_f16x2 __attribute__((ext_vector_type(2)));
  const h2_f16x2 hv = __builtin_bit_cast(h2_f16x2, h);
  const h2_f16x2 lv = {(_Float16)(x - (float)hv.x), (_Float16)(y - (float)hv.y)};
  return __builtin_bit_cast(unsigned, lv);
#else
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
      : "=&v"(l) : "v"(h), "v"(x), "v"(y));
  return l;
#endif
}

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
