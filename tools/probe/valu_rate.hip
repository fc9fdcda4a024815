// Diagnostic: issue cost (shader clocks per wave64 instruction) of the VALU instructions on the attention kernel's softmax side,
// and whether one wave's VALU stream overlaps an MFMA chain (its own, or another wave's on the same SIMD).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/valu_rate.hip -o tools/bin/valu_rate && tools/bin/valu_rate
// One wave per probe unless stated; 8 independent register chains, 32 instructions per loop trip; s_memtime = shader clocks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { EXP, FMA, PKFMA, PKADD, MAX3, CVTPK, MIXLO, MIXHI, CVT32, SWAP, LDEXP, MFMA, MFMA_PKFMA7, MFMA_DEP, MFMA_DEP_PKFMA7, MFMA_DEP_EXP4, NPROBE };
static const char* NAMES[NPROBE] = {"v_exp_f32", "v_fma_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_max3_f32", "v_cvt_pk_f16_f32", "v_fma_mixlo_f16", "v_fma_mixhi_f16",
                                    "v_cvt_f32_f16", "v_permlane32_swap", "v_ldexp_f32", "mfma 32x32x16 f16 (4 independent accumulators)",
                                    "1 mfma (independent) + 7 v_pk_fma_f32, per group", "mfma dependent chain (one accumulator)",
                                    "1 mfma (dependent chain) + 7 v_pk_fma_f32, per group", "1 mfma (dependent chain) + 4 v_exp_f32, per group"};

template <int P>
__global__ __launch_bounds__(64) void probe(unsigned long long* out, float* sink, int iters) {
  float x[8];
  f32x2 y[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f32x2{x[i], x[i] + 0.5f}; u[i] = threadIdx.x * 77u + i; }
  f32x16 c[4] = {{0}, {0}, {0}, {0}};
  unsigned h = threadIdx.x * 2654435761u;
  unsigned ua[4], ub[4];
  for (int i = 0; i < 4; ++i) { h = h * 1664525u + 1013904223u; ua[i] = h & 0x3bff3bffu; h = h * 1664525u + 1013904223u; ub[i] = h & 0x3bff3bffu; }
  f16x8 a, b;
  __builtin_memcpy(&a, ua, 16);
  __builtin_memcpy(&b, ub, 16);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (P == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
      REP32(X)
#undef X
    } else if constexpr (P == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
      REP32(X)
#undef X
    } else if constexpr (P == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y[i]));
      REP32(X)
#undef X
    } else if constexpr (P == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(y[i]));
      REP32(X)
#undef X
    } else if constexpr (P == MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
      REP32(X)
#undef X
    } else if constexpr (P == CVTPK) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(x[i]));
      REP32(X)
#undef X
    } else if constexpr (P == MIXLO) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(x[i]));
      REP32(X)
#undef X
    } else if constexpr (P == MIXHI) {
#define X(i) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(x[i]));
      REP32(X)
#undef X
    } else if constexpr (P == CVT32) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(u[i]));
      REP32(X)
#undef X
    } else if constexpr (P == SWAP) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 4) & 7]));
      REP8(X) REP8(X)
#undef X
    } else if constexpr (P == LDEXP) {
#define X(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(u[i]));
      REP32(X)
#undef X
    } else if constexpr (P == MFMA) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[k & 3], 0, 0, 0);
    } else if constexpr (P == MFMA_PKFMA7 || P == MFMA_DEP_PKFMA7) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c[P == MFMA_PKFMA7 ? k : 0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[P == MFMA_PKFMA7 ? k : 0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y[i]));
        X(0) X(1) X(2) X(3) X(4) X(5) X(6)
#undef X
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (P == MFMA_DEP) {
#pragma unroll
      for (int k = 0; k < 8; ++k) c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[0], 0, 0, 0);
    } else if constexpr (P == MFMA_DEP_EXP4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
        X(0) X(1) X(2) X(3)
#undef X
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y + (float)u[i];
  for (int k = 0; k < 4; ++k)
    for (int r = 0; r < 16; ++r) s += c[k][r];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// one trip of 32 (16 for the swap) instructions of probe P on the 8 register chains
template <int P>
__device__ __forceinline__ void trip(float (&x)[8], f32x2 (&y)[8], unsigned (&u)[8]) {
  if constexpr (P == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    REP32(X)
#undef X
  } else if constexpr (P == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
    REP32(X)
#undef X
  } else if constexpr (P == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y[i]));
    REP32(X)
#undef X
  } else if constexpr (P == PKADD) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(y[i]));
    REP32(X)
#undef X
  } else if constexpr (P == MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
    REP32(X)
#undef X
  } else if constexpr (P == CVTPK) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(x[i]));
    REP32(X)
#undef X
  } else if constexpr (P == MIXLO) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(x[i]));
    REP32(X)
#undef X
  } else if constexpr (P == MIXHI) {
#define X(i) asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(x[i]));
    REP32(X)
#undef X
  } else if constexpr (P == CVT32) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(u[i]));
    REP32(X)
#undef X
  } else if constexpr (P == SWAP) {
#define X(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u[i]), "+v"(u[(i + 4) & 7]));
    REP8(X) REP8(X)
#undef X
  } else if constexpr (P == LDEXP) {
#define X(i) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(x[i]) : "v"(u[i]));
    REP32(X)
#undef X
  }
}

// two waves on ONE SIMD (a 512-thread workgroup places waves w and w + 4 on the same SIMD): wave 0 runs MFMAs back to back (two
// accumulators in turn: the pipe never drains), wave 4 the VALU stream of probe P; each is also timed alone.
// who: bit 0 = the MFMA wave runs, bit 1 = the VALU wave runs
template <int P>
__global__ __launch_bounds__(512) void pair(unsigned long long* out, float* sink, int iters, int who) {
  const int wave = threadIdx.x >> 6;
  float x[8];
  f32x2 y[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f32x2{x[i], x[i] + 0.5f}; u[i] = threadIdx.x * 77u + i; }
  f32x16 c0 = {0}, c1 = {0};
  unsigned ua[4] = {0x3c003c00u, 0x38003800u, 0x3c003800u, 0x34003c00u};
  f16x8 a;
  __builtin_memcpy(&a, ua, 16);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave == 0 && (who & 1)) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c1, 0, 0, 0);
      }
  } else if (wave == 4 && (who & 2)) {
    for (int it = 0; it < iters; ++it) trip<P>(x, y, u);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y + (float)u[i];
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
  if (s == 12345.678f) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
}

// ONE wave, ONE register chain: every instruction reads the result of the one before it (issue-to-issue latency of dependent work)
template <int P>
__global__ __launch_bounds__(64) void chain(unsigned long long* out, float* sink, int iters) {
  float x = threadIdx.x * 1e-3f;
  f32x2 y = {x, x + 0.5f};
  unsigned u = threadIdx.x * 77u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      if constexpr (P == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
      else if constexpr (P == FMA) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x));
      else if constexpr (P == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y));
      else if constexpr (P == PKADD) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(y));
      else if constexpr (P == MAX3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x));
      else if constexpr (P == CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(x));
      else if constexpr (P == MIXLO) asm volatile("v_fma_mixlo_f16 %0, %0, -1.0, %1 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u) : "v"(x));
      else if constexpr (P == CVT32) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x));
      else if constexpr (P == SWAP) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(u), "+v"(x));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (x + y.x + y.y + (float)u == 12345.678f) sink[0] = x;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int P>
static void run_chain(unsigned long long* d_out, float* d_sink, int iters) {
  unsigned long long h = 0;
  hipLaunchKernelGGL(chain<P>, dim3(1), dim3(64), 0, 0, d_out, d_sink, 16);
  hipLaunchKernelGGL(chain<P>, dim3(1), dim3(64), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
  printf("dependent chain of %-20s %8.2f clocks each\n", NAMES[P], (double)h / ((double)iters * 32));
}

// ONE wave: a dependent MFMA chain with 8 instructions of probe P behind every MFMA (does the wave's own VALU stream run under its MFMA?)
template <int P>
__device__ __forceinline__ void eight(float (&x)[8], f32x2 (&y)[8], unsigned (&u)[8]) {
  if constexpr (P == EXP) {
#define X(i) asm volatile("v_exp_f32 %0, %0" : "+v"(x[i]));
    REP8(X)
#undef X
  } else if constexpr (P == FMA) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(x[i]));
    REP8(X)
#undef X
  } else if constexpr (P == PKFMA) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(y[i]));
    REP8(X)
#undef X
  } else if constexpr (P == MAX3) {
#define X(i) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x[i]));
    REP8(X)
#undef X
  } else if constexpr (P == CVTPK) {
#define X(i) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "=v"(u[i]) : "v"(x[i]));
    REP8(X)
#undef X
  } else if constexpr (P == MIXLO) {
#define X(i) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(x[i]));
    REP8(X)
#undef X
  } else if constexpr (P == CVT32) {
#define X(i) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(u[i]));
    REP8(X)
#undef X
  }
}
template <int P>
__global__ __launch_bounds__(64) void group(unsigned long long* out, float* sink, int iters) {
  float x[8];
  f32x2 y[8];
  unsigned u[8];
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3f + i; y[i] = f32x2{x[i], x[i] + 0.5f}; u[i] = threadIdx.x * 77u + i; }
  f32x16 c = {0};
  unsigned ua[4] = {0x3c003c00u, 0x38003800u, 0x3c003800u, 0x34003c00u};
  f16x8 a;
  __builtin_memcpy(&a, ua, 16);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, c, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      eight<P>(x, y, u);
      __builtin_amdgcn_sched_barrier(0);
    }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i] + y[i].x + y[i].y + (float)u[i];
  for (int r = 0; r < 16; ++r) s += c[r];
  if (s == 12345.678f) sink[0] = s;
  if (threadIdx.x == 0) out[0] = t1 - t0;
}
template <int P>
static void run_group(unsigned long long* d_out, float* d_sink, int iters) {
  unsigned long long h = 0;
  hipLaunchKernelGGL(group<P>, dim3(1), dim3(64), 0, 0, d_out, d_sink, 16);
  hipLaunchKernelGGL(group<P>, dim3(1), dim3(64), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
  printf("one wave: 1 mfma (dependent chain) + 8 x %-20s %8.2f clocks per group\n", NAMES[P], (double)h / ((double)iters * 4));
}

template <int P>
static void run_pair(unsigned long long* d_out, float* d_sink, int iters, int per_trip) {
  double mf[4] = {0}, va[4] = {0};
  for (int who = 1; who <= 3; ++who) {
    unsigned long long h[8] = {0};
    hipMemset(d_out, 0, 64);
    hipLaunchKernelGGL(pair<P>, dim3(1), dim3(512), 0, 0, d_out, d_sink, iters, who);
    hipDeviceSynchronize();
    hipMemcpy(h, d_out, 64, hipMemcpyDeviceToHost);
    mf[who] = (double)h[0] / (iters * 8.0);
    va[who] = (double)h[4] / ((double)iters * per_trip);
  }
  // per 32 clocks of MFMA: how many of the VALU instructions go through next to it, as a fraction of the rate alone
  printf("%-20s alone %6.2f clocks; next to a full MFMA pipe %6.2f clocks (mfma %6.2f alone, %6.2f next to it): %3.0f %% of its own rate\n", NAMES[P], va[2], va[3], mf[1], mf[3],
         100.0 * va[2] / va[3]);
}

template <int P>
static void run(unsigned long long* d_out, float* d_sink, int iters, int per_trip) {
  unsigned long long h = 0;
  hipLaunchKernelGGL(probe<P>, dim3(1), dim3(64), 0, 0, d_out, d_sink, 16);            // warm-up (instruction cache)
  hipLaunchKernelGGL(probe<P>, dim3(1), dim3(64), 0, 0, d_out, d_sink, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, d_out, 8, hipMemcpyDeviceToHost);
  printf("%-58s %8.2f clocks each (%d per trip)\n", NAMES[P], (double)h / ((double)iters * per_trip), per_trip);
}

int main() {
  unsigned long long* d_out;
  float* d_sink;
  hipMalloc(&d_out, 64 * 8);
  hipMalloc(&d_sink, 64);
  const int iters = 2000;
  run<EXP>(d_out, d_sink, iters, 32);
  run<FMA>(d_out, d_sink, iters, 32);
  run<PKFMA>(d_out, d_sink, iters, 32);
  run<PKADD>(d_out, d_sink, iters, 32);
  run<MAX3>(d_out, d_sink, iters, 32);
  run<CVTPK>(d_out, d_sink, iters, 32);
  run<MIXLO>(d_out, d_sink, iters, 32);
  run<MIXHI>(d_out, d_sink, iters, 32);
  run<CVT32>(d_out, d_sink, iters, 32);
  run<SWAP>(d_out, d_sink, iters, 16);
  run<LDEXP>(d_out, d_sink, iters, 32);
  run<MFMA>(d_out, d_sink, iters, 8);
  run<MFMA_PKFMA7>(d_out, d_sink, iters, 4);
  run<MFMA_DEP>(d_out, d_sink, iters, 8);
  run<MFMA_DEP_PKFMA7>(d_out, d_sink, iters, 4);
  run<MFMA_DEP_EXP4>(d_out, d_sink, iters, 4);
  run_chain<EXP>(d_out, d_sink, iters);
  run_chain<FMA>(d_out, d_sink, iters);
  run_chain<PKFMA>(d_out, d_sink, iters);
  run_chain<PKADD>(d_out, d_sink, iters);
  run_chain<MAX3>(d_out, d_sink, iters);
  run_chain<CVTPK>(d_out, d_sink, iters);
  run_chain<MIXLO>(d_out, d_sink, iters);
  run_chain<CVT32>(d_out, d_sink, iters);
  run_chain<SWAP>(d_out, d_sink, iters);
  run_group<EXP>(d_out, d_sink, iters);
  run_group<FMA>(d_out, d_sink, iters);
  run_group<PKFMA>(d_out, d_sink, iters);
  run_group<MAX3>(d_out, d_sink, iters);
  run_group<CVTPK>(d_out, d_sink, iters);
  run_group<MIXLO>(d_out, d_sink, iters);
  run_group<CVT32>(d_out, d_sink, iters);
  printf("-- two waves on one SIMD: the VALU stream of wave 4 next to back-to-back MFMAs of wave 0\n");
  run_pair<EXP>(d_out, d_sink, iters, 32);
  run_pair<FMA>(d_out, d_sink, iters, 32);
  run_pair<PKFMA>(d_out, d_sink, iters, 32);
  run_pair<PKADD>(d_out, d_sink, iters, 32);
  run_pair<MAX3>(d_out, d_sink, iters, 32);
  run_pair<CVTPK>(d_out, d_sink, iters, 32);
  run_pair<MIXLO>(d_out, d_sink, iters, 32);
  run_pair<CVT32>(d_out, d_sink, iters, 32);
  run_pair<SWAP>(d_out, d_sink, iters, 16);
  run_pair<LDEXP>(d_out, d_sink, iters, 32);
  return 0;
}
