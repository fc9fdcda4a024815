// ldmk_post: the launch that FOLLOWS a GEMM in the small-batch (latency-bound) route of the UNet.
//
// At batch 1 (the reference's shipped talking-face mode, progressive_sampling_difftalk.py:282-317) a DDIM step is a chain
// of ~400 dependent launches of 3-14 us each (profiles/r02_layers32_b1.txt): split-K GEMM, its reduce kernel, GroupNorm
// finalize, GroupNorm apply, LayerNorm statistics ...  Every one of those is a full pass over a tensor of 40-650 KB that
// pays a launch ramp and one or two memory round trips.  This kernel folds what sits between two GEMMs into ONE launch:
//
//     v      = alpha * sum_k slab_k  + bias + per-sample vector + residual        (the split-K reduce + GEMM epilogue)
//     raw    = v  (GEGLU: value * gelu(gate) of the packed 32-column pairs)        (optional store: residual stream / skip)
//     normed = GroupNorm(32)[+SiLU] of the channel concat (raw | x1), or LayerNorm(raw) * gamma + beta   (optional)
//
// so a ResBlock is  conv1 -> post -> conv2 -> post  and a transformer block  proj_in -> post -> qkv -> attention ->
// to_out -> post -> GEGLU proj -> [post] -> ff.net.2*proj_out -> post  (DESIGN.md section 12).
//   GroupNorm mode: one workgroup per (sample, group); the group's hw x cpg values are reduced, kept in LDS, their mean and
//     centred second moment taken in two passes (no E[x^2] - E[x]^2 cancellation), normalised and written -- the statistics
//     never exist in memory and groups may straddle the concat seam (openaimodel.py:736: 320 + 160 channels = 15 per group).
//   LayerNorm mode: one wave per row (float4 lanes), two-pass in registers.
//   plain / GEGLU mode: elementwise.
// All sums run in a fixed order: results are bitwise reproducible (hipGraph replay == eager launches).
#include "ldmk_common.h"
#include <limits.h>

namespace ldmk {

constexpr int POST_LDS_FLOATS = 15 * 1024;      // 60 KB value cache of the GroupNorm mode (two workgroups per CU)

// Slab sums with many loads in flight.  The slab order of the additions is what fixes the result; the LOADS of up to
// POST_SB slabs (x POST_EB elements in the GroupNorm mode) are issued together, because a launch of this kernel is a
// chain of memory round trips and one load per round trip would make it nslab of them (first version: 23 us per call).
// Slabs past nslab contribute +0.0f (x + 0 == x).
constexpr int POST_SB = 8;      // slabs per batch
constexpr int POST_EB = 4;      // elements per batch (GroupNorm mode)

__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ base, int nslab, long long stride) {
  float4 a = *reinterpret_cast<const float4*>(base);
  for (int k = 1; k < nslab; k += POST_SB) {
    float4 t[POST_SB];
#pragma unroll
    for (int u = 0; u < POST_SB; ++u)
      t[u] = k + u < nslab ? *reinterpret_cast<const float4*>(base + (long long)(k + u) * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < POST_SB; ++u) { a.x += t[u].x; a.y += t[u].y; a.z += t[u].z; a.w += t[u].w; }
  }
  return a;
}

// ---- GroupNorm mode: grid (groups, samples), 1024 threads: a (sample, group) of hw x cpg values is 1-20 values per thread
constexpr int POST_GN_THREADS = 1024;
__global__ __launch_bounds__(POST_GN_THREADS) void post_gn_kernel(const ldmk_post_args p) {
  extern __shared__ float cache[];                       // [min(items, cap)] values of this (sample, group)
  __shared__ float red[2][POST_GN_THREADS / 64];
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.N + p.c1;
  const int cpg = C / p.groups;
  const int hw = p.rows_per_sample;
  const int items = hw * cpg;
  const int cap = p.gn_cache_floats;
  const int cbase = g * cpg;
  const long long row0 = (long long)n * hw;
  const float* __restrict__ bvec = p.batch_vec ? p.batch_vec + (long long)n * p.batch_vec_ld : nullptr;
  // pass 1: values -> raw_out (source part) and the LDS cache; running sum.  POST_EB elements x POST_SB slabs in flight.
  float sum = 0.f;
  for (int i0 = tid; i0 < items; i0 += POST_GN_THREADS * POST_EB) {
    float v[POST_EB];
    long long off[POST_EB];                              // element offset in the source / in x1 (clamped: loads are unconditional)
    int col[POST_EB], rowi[POST_EB];
    bool in_src[POST_EB], live[POST_EB];
#pragma unroll
    for (int e = 0; e < POST_EB; ++e) {
      const int i = min(i0 + e * POST_GN_THREADS, items - 1);
      const int pix = i / cpg, c = cbase + (i - pix * cpg);
      const long long row = row0 + pix;
      live[e] = i0 + e * POST_GN_THREADS < items;
      in_src[e] = c < p.N;
      col[e] = c;
      rowi[e] = (int)row;
      off[e] = in_src[e] ? row * p.N + c : row * p.c1 + (c - p.N);
      v[e] = (in_src[e] ? p.src : p.x1)[off[e]];
    }
    for (int k = 1; k < p.nslab; k += POST_SB) {
      float t[POST_SB][POST_EB];
#pragma unroll
      for (int u = 0; u < POST_SB; ++u) {
        const long long so = (long long)min(k + u, p.nslab - 1) * p.slab_stride;     // clamped: the load is unconditional
#pragma unroll
        for (int e = 0; e < POST_EB; ++e) t[u][e] = p.src[so + (in_src[e] ? off[e] : 0)];
      }
#pragma unroll
      for (int u = 0; u < POST_SB; ++u)
#pragma unroll
        for (int e = 0; e < POST_EB; ++e) v[e] += (k + u < p.nslab && in_src[e]) ? t[u][e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < POST_EB; ++e) {
      const int i = i0 + e * POST_GN_THREADS;
      if (in_src[e]) {                                   // source part: epilogue terms, same association as the GEMM epilogues
        float s = v[e] * p.alpha;
        if (p.bias) s += p.bias[col[e]];
        if (bvec) s += bvec[col[e]];
        if (p.residual) s += p.residual[(long long)rowi[e] * p.ldr + col[e]];
        if (p.raw_out && live[e]) p.raw_out[(long long)rowi[e] * p.ld_raw + col[e]] = s;
        v[e] = s;
      }
      if (live[e]) {
        if (i < cap) cache[i] = v[e];
        sum += v[e];
      }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[0][wave] = sum;
  __syncthreads();
  float tot = 0.f;
#pragma unroll
  for (int w = 0; w < POST_GN_THREADS / 64; ++w) tot += red[0][w];
  const float mean = tot / (float)items;
  // values beyond the cache are read back (this thread stored them itself) or, for a plain source, simply read again
  auto reload = [&](int i) -> float {
    const int pix = i / cpg, c = cbase + (i - pix * cpg);
    const long long row = row0 + pix;
    if (c >= p.N) return p.x1[row * p.c1 + (c - p.N)];
    return p.raw_out ? p.raw_out[row * p.ld_raw + c] : p.src[row * p.N + c];
  };
  // pass 2: centred second moment
  float sq = 0.f;
  for (int i = tid; i < items; i += POST_GN_THREADS) {
    const float d = (i < cap ? cache[i] : reload(i)) - mean;
    sq = fmaf(d, d, sq);
  }
  sq = wave_sum(sq);
  if (lane == 0) red[1][wave] = sq;
  __syncthreads();
  float tq = 0.f;
#pragma unroll
  for (int w = 0; w < POST_GN_THREADS / 64; ++w) tq += red[1][w];
  const float rstd = 1.0f / sqrtf(tq / (float)items + p.eps);
  // pass 3: normalise (+SiLU) -> norm_out [M][C]
  for (int i = tid; i < items; i += POST_GN_THREADS) {
    const int pix = i / cpg, c = cbase + (i - pix * cpg);
    const float v = i < cap ? cache[i] : reload(i);
    float y = fmaf((v - mean) * rstd, p.gamma[c], p.beta[c]);
    if (p.silu) y = silu_f(y);
    p.norm_out[(row0 + pix) * p.ld_norm + c] = y;
  }
}

// ---- LayerNorm mode: one wave per row, float4 lanes (N % 4 == 0, N <= 2048); one single-wave workgroup per row, so that
// the rows of a 64-row problem spread over 64 CUs
__global__ __launch_bounds__(64) void post_ln_kernel(const ldmk_post_args p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  if (row >= p.M) return;
  const int n4 = p.N >> 2;
  const int sample = row / p.rows_per_sample;
  float4 v[8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c4 = lane + 64 * j;
    v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c4 < n4) {
      const long long o = (long long)row * p.N + 4 * c4;
      float4 a = slab_sum4(p.src + o, p.nslab, p.slab_stride);
      a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
      if (p.bias) {
        const float4 t = *reinterpret_cast<const float4*>(p.bias + 4 * c4);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      if (p.batch_vec) {
        const float4 t = *reinterpret_cast<const float4*>(p.batch_vec + (long long)sample * p.batch_vec_ld + 4 * c4);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      if (p.residual) {
        const float4 t = *reinterpret_cast<const float4*>(p.residual + (long long)row * p.ldr + 4 * c4);
        a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      }
      if (p.raw_out) *reinterpret_cast<float4*>(p.raw_out + (long long)row * p.ld_raw + 4 * c4) = a;
      v[j] = a;
      s += (a.x + a.y) + (a.z + a.w);
    }
  }
  s = wave_sum(s);
  const float mean = s / (float)p.N;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (lane + 64 * j < n4) {
      const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      q = fmaf(a, a, q); q = fmaf(b, b, q); q = fmaf(c, c, q); q = fmaf(d, d, q);
    }
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)p.N + p.eps);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c4 = lane + 64 * j;
    if (c4 < n4) {
      const float4 ga = *reinterpret_cast<const float4*>(p.gamma + 4 * c4);
      const float4 be = *reinterpret_cast<const float4*>(p.beta + 4 * c4);
      float4 y;
      y.x = fmaf((v[j].x - mean) * rstd, ga.x, be.x); y.y = fmaf((v[j].y - mean) * rstd, ga.y, be.y);
      y.z = fmaf((v[j].z - mean) * rstd, ga.z, be.z); y.w = fmaf((v[j].w - mean) * rstd, ga.w, be.w);
      *reinterpret_cast<float4*>(p.norm_out + (long long)row * p.ld_norm + 4 * c4) = y;
    }
  }
}

// ---- plain mode: elementwise reduce + epilogue -> raw_out (float4)
__global__ __launch_bounds__(256) void post_plain_kernel(const ldmk_post_args p) {
  const int n4 = p.N >> 2;
  const long long total = (long long)p.M * n4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int row = (int)(i / n4), c = (int)(i - (long long)row * n4) * 4;
    const long long o = (long long)row * p.N + c;
    float4 a = slab_sum4(p.src + o, p.nslab, p.slab_stride);
    a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
    if (p.bias) {
      const float4 t = *reinterpret_cast<const float4*>(p.bias + c);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    if (p.batch_vec) {
      const float4 t = *reinterpret_cast<const float4*>(p.batch_vec + (long long)(row / p.rows_per_sample) * p.batch_vec_ld + c);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    if (p.residual) {
      const float4 t = *reinterpret_cast<const float4*>(p.residual + (long long)row * p.ldr + c);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
    }
    *reinterpret_cast<float4*>(p.raw_out + (long long)row * p.ld_raw + c) = a;
  }
}

// ---- GEGLU mode: src columns are the packed (value | gate) 32-column pairs of ldmk_igemm's GEGLU layout;
// raw_out[M][N/2] = (value + bias_v) * gelu(gate + bias_g)      (attention.py:37-50)
__global__ __launch_bounds__(256) void post_geglu_kernel(const ldmk_post_args p) {
  const int half = p.N >> 1;
  const int h4 = half >> 2;
  const long long total = (long long)p.M * h4;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int row = (int)(i / h4), c = (int)(i - (long long)row * h4) * 4;      // output column (4 of them: same 32-block)
    const int cv = ((c >> 5) << 6) + (c & 31), cg = cv + 32;                      // packed source columns
    const long long ov = (long long)row * p.N + cv, og = (long long)row * p.N + cg;
    float4 a = slab_sum4(p.src + ov, p.nslab, p.slab_stride), b = slab_sum4(p.src + og, p.nslab, p.slab_stride);
    a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
    b.x *= p.alpha; b.y *= p.alpha; b.z *= p.alpha; b.w *= p.alpha;
    if (p.bias) {
      const float4 t = *reinterpret_cast<const float4*>(p.bias + cv), u = *reinterpret_cast<const float4*>(p.bias + cg);
      a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
      b.x += u.x; b.y += u.y; b.z += u.z; b.w += u.w;
    }
    float4 y;
    y.x = a.x * gelu_erf_f(b.x); y.y = a.y * gelu_erf_f(b.y); y.z = a.z * gelu_erf_f(b.z); y.w = a.w * gelu_erf_f(b.w);
    *reinterpret_cast<float4*>(p.raw_out + (long long)row * p.ld_raw + c) = y;
  }
}

}  // namespace ldmk

extern "C" int ldmk_post(const ldmk_post_args* args, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(args != nullptr, "ldmk_post: null args");
  ldmk_post_args p = *args;
  if (p.alpha == 0.f) p.alpha = 1.f;
  hipStream_t st = (hipStream_t)stream;
  LDMK_REQUIRE(p.src && p.M > 0 && p.N > 0 && p.nslab >= 1 && p.rows_per_sample > 0 && p.M % p.rows_per_sample == 0,
               "ldmk_post: bad source (M=%d N=%d nslab=%d rows_per_sample=%d)", p.M, p.N, p.nslab, p.rows_per_sample);
  LDMK_REQUIRE(p.nslab == 1 || p.slab_stride >= (long long)p.M * p.N, "ldmk_post: slab_stride smaller than one slab");
  LDMK_REQUIRE(p.N % 4 == 0, "ldmk_post: N must be a multiple of 4");
  LDMK_REQUIRE(!p.batch_vec || p.batch_vec_ld >= p.N, "ldmk_post: batch_vec_ld < N");
  LDMK_REQUIRE(!p.residual || p.ldr >= p.N, "ldmk_post: residual row stride < N");
  if (p.geglu) {
    LDMK_REQUIRE(p.norm == LDMK_POST_NONE && p.raw_out && !p.residual && !p.batch_vec && p.N % 64 == 0 && p.ld_raw >= p.N / 2,
                 "ldmk_post: GEGLU takes packed (value, gate) pairs (N %% 64 == 0) to raw_out[M][N/2], no norm / residual");
    long long g = ((long long)p.M * (p.N / 8) + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(post_geglu_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
    return check_launch("ldmk_post(geglu)");
  }
  LDMK_REQUIRE(!p.raw_out || p.ld_raw >= p.N, "ldmk_post: raw_out row stride < N");
  if (p.norm == LDMK_POST_NONE) {
    LDMK_REQUIRE(p.raw_out != nullptr, "ldmk_post: nothing to write (raw_out and norm both off)");
    long long g = ((long long)p.M * (p.N / 4) + 255) / 256;
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(post_plain_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
    return check_launch("ldmk_post(plain)");
  }
  LDMK_REQUIRE(p.gamma && p.beta && p.norm_out, "ldmk_post: norm needs gamma, beta and norm_out");
  if (p.norm == LDMK_POST_LAYERNORM) {
    LDMK_REQUIRE(p.N <= 2048 && !p.x1 && p.c1 == 0 && p.ld_norm >= p.N, "ldmk_post: LayerNorm rows of at most 2048 floats, no concat");
    hipLaunchKernelGGL(post_ln_kernel, dim3(p.M), dim3(64), 0, st, p);
    return check_launch("ldmk_post(layernorm)");
  }
  LDMK_REQUIRE(p.norm == LDMK_POST_GROUPNORM, "ldmk_post: unknown norm %d", p.norm);
  const int C = p.N + p.c1;
  LDMK_REQUIRE((p.c1 == 0) == (p.x1 == nullptr) && p.c1 >= 0, "ldmk_post: x1 / c1 mismatch");
  LDMK_REQUIRE(p.groups > 0 && C % p.groups == 0 && p.ld_norm >= C, "ldmk_post: C=%d groups=%d ld_norm=%d", C, p.groups, p.ld_norm);
  const long long items = (long long)p.rows_per_sample * (C / p.groups);
  LDMK_REQUIRE(items < (1LL << 30), "ldmk_post: group too large");
  int cache = (int)(items < POST_LDS_FLOATS ? items : POST_LDS_FLOATS);
  // values beyond the cache are read back in the later passes: from raw_out, or from the source itself when that is a plain
  // tensor (one slab, no epilogue terms)
  const bool plain_src = p.nslab == 1 && !p.bias && !p.batch_vec && !p.residual && p.alpha == 1.0f;
  LDMK_REQUIRE(items <= POST_LDS_FLOATS || p.raw_out || plain_src,
               "ldmk_post: groups of more than %d values need raw_out (or a plain source)", POST_LDS_FLOATS);
  p.gn_cache_floats = cache;
  hipLaunchKernelGGL(post_gn_kernel, dim3(p.groups, p.M / p.rows_per_sample), dim3(POST_GN_THREADS), (size_t)cache * sizeof(float), st, p);
  return check_launch("ldmk_post(groupnorm)");
}
