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

constexpr int POST_LDS_FLOATS = 15 * 1024;      // 60 KB value cache of the GroupNorm mode

// A launch of these kernels is a chain of memory round trips (~1.5-2.5 us each when the slabs were written by CUs of
// another XCD), so every load that does not depend on loaded data is ISSUED before the first wait: the slabs of a batch,
// the epilogue operands (bias, per-sample vector, residual) and the norm's gamma / beta.  First version: one load per
// round trip, 23 us per call; second: slab loads batched but the epilogue and gamma / beta loads still one by one, 12 us.
// The slab ORDER of the additions is what fixes the result (slabs past nslab contribute +0.0f: x + 0 == x).
constexpr int POST_SB = 8;      // slabs per batch
constexpr int POST_EB = 4;      // elements per batch (GroupNorm mode)

__device__ __forceinline__ float4 f4_add(float4 a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; return a; }
__device__ __forceinline__ float4 f4_ld(const float* p_) { return *reinterpret_cast<const float4*>(p_); }
constexpr float4 F4_ZERO = {0.f, 0.f, 0.f, 0.f};

// remaining slabs 1 .. nslab-1 added to `a` (slab 0), POST_SB loads in flight
__device__ __forceinline__ float4 slab_rest4(float4 a, const float* __restrict__ base, int nslab, long long stride) {
  for (int k = 1; k < nslab; k += POST_SB) {
    float4 t[POST_SB];
#pragma unroll
    for (int u = 0; u < POST_SB; ++u) t[u] = f4_ld(base + (long long)min(k + u, nslab - 1) * stride);     // clamped: unconditional
#pragma unroll
    for (int u = 0; u < POST_SB; ++u) a = f4_add(a, k + u < nslab ? t[u] : F4_ZERO);
  }
  return a;
}

// ---- GroupNorm mode: grid (groups, samples), 1024 threads: a (sample, group) of hw x cpg values is 1-20 values per thread
constexpr int POST_GN_THREADS = 1024;
__global__ __launch_bounds__(POST_GN_THREADS) void post_gn_kernel(const ldmk_post_args p) {
  extern __shared__ float cache[];                       // [min(items, cap)] values of this (sample, group)
  __shared__ float red[2][POST_GN_THREADS / 64];
  __shared__ float gb[2][64];                            // this group's gamma / beta (cpg <= 64)
  const int g = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int C = p.N + p.c1;
  const int cpg = C / p.groups;
  const int hw = p.rows_per_sample;
  const int items = hw * cpg;
  const int cap = p.gn_cache_floats;
  const int cbase = g * cpg;
  const long long row0 = (long long)n * hw;
  const float* __restrict__ bvec = p.batch_vec ? p.batch_vec + (long long)n * p.batch_vec_ld : nullptr;
  if (tid < cpg) { gb[0][tid] = p.gamma[cbase + tid]; gb[1][tid] = p.beta[cbase + tid]; }
  // pass 1: values -> raw_out (source part) and the LDS cache; running sum.  POST_EB elements x POST_SB slabs in flight.
  float sum = 0.f;
  for (int i0 = tid; i0 < items; i0 += POST_GN_THREADS * POST_EB) {
    float v[POST_EB], eb[POST_EB], ev[POST_EB], er[POST_EB];
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
      col[e] = min(c, p.N - 1);
      rowi[e] = (int)row;
      off[e] = in_src[e] ? row * p.N + c : row * p.c1 + (c - p.N);
    }
#pragma unroll
    for (int e = 0; e < POST_EB; ++e) v[e] = (in_src[e] ? p.src : p.x1)[off[e]];
    // epilogue operands: requested now, used after the slab sums
#pragma unroll
    for (int e = 0; e < POST_EB; ++e) {
      eb[e] = p.bias ? p.bias[col[e]] : 0.f;
      ev[e] = bvec ? bvec[col[e]] : 0.f;
      er[e] = p.residual ? p.residual[(long long)rowi[e] * p.ldr + col[e]] : 0.f;
    }
    for (int k = 1; k < p.nslab; k += POST_SB) {
      float t[POST_SB][POST_EB];
#pragma unroll
      for (int u = 0; u < POST_SB; ++u) {
        const long long so = (long long)min(k + u, p.nslab - 1) * p.slab_stride;
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
        if (p.bias) s += eb[e];
        if (bvec) s += ev[e];
        if (p.residual) s += er[e];
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
  // pass 3: normalise (+SiLU) -> norm_out [M][C]  (values and gamma / beta from LDS: no load in this loop)
  for (int i = tid; i < items; i += POST_GN_THREADS) {
    const int pix = i / cpg, cc = i - pix * cpg;
    const float v = i < cap ? cache[i] : reload(i);
    float y = fmaf((v - mean) * rstd, gb[0][cc], gb[1][cc]);
    if (p.silu) y = silu_f(y);
    p.norm_out[(row0 + pix) * p.ld_norm + cbase + cc] = y;
  }
}

// ---- GroupNorm mode for LARGE images (rows_per_sample >= 256), two launches behind one call.
// One workgroup per (sample, group) is 32 workgroups per sample walking 5-15 K values each through 4-byte strided
// accesses: 25-53 us per call at 32x32 (profiles/r03: the 1024-pixel levels), against 7.5 us at 8x8.  Here both launches
// are tiled over ROWS, use float4 accesses along the channel axis and fill the chip:
//   stats:  workgroup = GS_R rows x all channels: slab sum + epilogue -> raw_out, the tile kept in LDS, and per (tile, group) the
//           pair (mean, centred sum of squares), two-pass inside the tile;
//   apply:  workgroup = GS_R rows: folds the (tile, group) pairs of its sample with Chan's update in tile order (double), then
//           y = (v - mean) rstd gamma + beta [SiLU] over its rows.
constexpr int GS_R = 8;
__global__ __launch_bounds__(256) void post_gnstat_kernel(const ldmk_post_args p, float* __restrict__ rec) {
  extern __shared__ float tile[];                       // [GS_R][C]
  const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int C = p.N + p.c1, cpg = C / p.groups, hw = p.rows_per_sample;
  const int r0 = chunk * GS_R, nr = min(GS_R, hw - r0);
  const long long row0 = (long long)n * hw + r0;
  const int n4 = p.N >> 2, c4 = C >> 2;                  // (N % 4 == 0 and c1 % 4 == 0: a float4 never straddles the seam)
  const float* __restrict__ bvec = p.batch_vec ? p.batch_vec + (long long)n * p.batch_vec_ld : nullptr;
  const bool pending = p.raw_out != nullptr;
  for (int i = tid; i < nr * c4; i += 256) {
    const int r = i / c4, q = i - r * c4;
    const long long row = row0 + r;
    float4 a;
    if (q < n4) {
      const int c = 4 * q;
      const float* base = p.src + row * p.N + c;
      a = f4_ld(base);
      const float4 eb = p.bias ? f4_ld(p.bias + c) : F4_ZERO;
      const float4 ev = bvec ? f4_ld(bvec + c) : F4_ZERO;
      const float4 er = p.residual ? f4_ld(p.residual + row * p.ldr + c) : F4_ZERO;
      a = slab_rest4(a, base, p.nslab, p.slab_stride);
      a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
      if (p.bias) a = f4_add(a, eb);
      if (bvec) a = f4_add(a, ev);
      if (p.residual) a = f4_add(a, er);
      if (pending) *reinterpret_cast<float4*>(p.raw_out + row * p.ld_raw + c) = a;
    } else {
      a = f4_ld(p.x1 + row * p.c1 + 4 * (q - n4));
    }
    *reinterpret_cast<float4*>(tile + r * C + 4 * q) = a;
  }
  __syncthreads();
  // 8 lanes per group: mean, then centred sum of squares, of the tile's nr x cpg values of the group
  const int g = tid >> 3, l8 = tid & 7;
  const int cnt = nr * cpg;
  float s = 0.f;
  for (int i = l8; i < cnt; i += 8) { const int r = i / cpg; s += tile[r * C + g * cpg + (i - r * cpg)]; }
  s += __shfl_xor(s, 4, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 1, 64);
  const float mean = s / (float)cnt;
  float m2 = 0.f;
  for (int i = l8; i < cnt; i += 8) { const int r = i / cpg; const float d = tile[r * C + g * cpg + (i - r * cpg)] - mean; m2 = fmaf(d, d, m2); }
  m2 += __shfl_xor(m2, 4, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 1, 64);
  if (l8 == 0) {
    float* d = rec + (((long long)n * gridDim.x + chunk) * 32 + g) * 2;
    d[0] = mean; d[1] = m2;
  }
}

__global__ __launch_bounds__(256) void post_gnapply_kernel(const ldmk_post_args p, const float* __restrict__ rec, int chunks) {
  __shared__ float gm[32], gr[32];
  const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
  const int C = p.N + p.c1, cpg = C / p.groups, hw = p.rows_per_sample;
  {
    // per group  mean = sum n_k mean_k / sum n_k,  M2 = sum [M2_k + n_k (mean_k - mean)^2]  in double; 8 lanes per group, each
    // walking every 8th tile with 8 independent loads in flight (first version: one serial loop with a dependent global load
    // per tile, 39 us at 128 tiles), partial sums folded in a fixed order
    const int g = tid >> 3, l8 = tid & 7;
    const float2* r2 = reinterpret_cast<const float2*>(rec) + (long long)n * chunks * 32 + g;
    double sw = 0.0, sn = 0.0;
    for (int k0 = l8; k0 < chunks; k0 += 64) {
      float2 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = r2[(long long)min(k0 + 8 * u, chunks - 1) * 32];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 8 * u;
        const double nk = k < chunks ? (double)(min(GS_R, hw - k * GS_R) * cpg) : 0.0;
        sw += nk * (double)t[u].x;
        sn += nk;
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { sw += __shfl_xor(sw, o, 64); sn += __shfl_xor(sn, o, 64); }
    const double mean = sw / sn;
    double m2 = 0.0;
    for (int k0 = l8; k0 < chunks; k0 += 64) {
      float2 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = r2[(long long)min(k0 + 8 * u, chunks - 1) * 32];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = k0 + 8 * u;
        if (k < chunks) {
          const double nk = (double)(min(GS_R, hw - k * GS_R) * cpg), dm = (double)t[u].x - mean;
          m2 += (double)t[u].y + nk * dm * dm;
        }
      }
    }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) m2 += __shfl_xor(m2, o, 64);
    if (l8 == 0) {
      gm[g] = (float)mean;
      gr[g] = (float)(1.0 / sqrt(m2 / sn + (double)p.eps));
    }
  }
  __syncthreads();
  const int r0 = chunk * GS_R, nr = min(GS_R, hw - r0);
  const long long row0 = (long long)n * hw + r0;
  const int n4 = p.N >> 2, c4 = C >> 2;
  const float* __restrict__ vsrc = p.raw_out ? p.raw_out : p.src;      // the stats launch stored pending values in raw_out
  const int ldv = p.raw_out ? p.ld_raw : p.N;
  for (int i = tid; i < nr * c4; i += 256) {
    const int r = i / c4, q = i - r * c4, c = 4 * q;
    const long long row = row0 + r;
    const float4 v = q < n4 ? f4_ld(vsrc + row * ldv + c) : f4_ld(p.x1 + row * p.c1 + (c - p.N));
    const float4 ga = f4_ld(p.gamma + c), be = f4_ld(p.beta + c);
    const float vv[4] = {v.x, v.y, v.z, v.w}, gg[4] = {ga.x, ga.y, ga.z, ga.w}, bb[4] = {be.x, be.y, be.z, be.w};
    float y[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int g = (c + e) / cpg;
      y[e] = fmaf((vv[e] - gm[g]) * gr[g], gg[e], bb[e]);
      if (p.silu) y[e] = silu_f(y[e]);
    }
    *reinterpret_cast<float4*>(p.norm_out + row * p.ld_norm + c) = make_float4(y[0], y[1], y[2], y[3]);
  }
}

// ---- LayerNorm mode: one single-wave workgroup per row (the rows of a 64-row problem spread over 64 CUs), float4 lanes,
// JN = ceil(N / 256) float4 per lane; two-pass statistics in registers
template <int JN>
__global__ __launch_bounds__(64) void post_ln_kernel(const ldmk_post_args p) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x;
  const int n4 = p.N >> 2;
  const int sample = row / p.rows_per_sample;
  float4 v[JN], eb[JN], ev[JN], er[JN], ga[JN], be[JN];
  const float* srow = p.src + (long long)row * p.N;
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    const int c = 4 * min(lane + 64 * j, n4 - 1);            // clamped: every load below is unconditional
    v[j] = f4_ld(srow + c);
    eb[j] = p.bias ? f4_ld(p.bias + c) : F4_ZERO;
    ev[j] = p.batch_vec ? f4_ld(p.batch_vec + (long long)sample * p.batch_vec_ld + c) : F4_ZERO;
    er[j] = p.residual ? f4_ld(p.residual + (long long)row * p.ldr + c) : F4_ZERO;
    ga[j] = f4_ld(p.gamma + c);
    be[j] = f4_ld(p.beta + c);
  }
  if (p.nslab > 1) {
#pragma unroll
    for (int j = 0; j < JN; ++j) v[j] = slab_rest4(v[j], srow + 4 * min(lane + 64 * j, n4 - 1), p.nslab, p.slab_stride);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    float4 a = v[j];
    a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
    if (p.bias) a = f4_add(a, eb[j]);
    if (p.batch_vec) a = f4_add(a, ev[j]);
    if (p.residual) a = f4_add(a, er[j]);
    v[j] = a;
    if (lane + 64 * j < n4) {
      if (p.raw_out) *reinterpret_cast<float4*>(p.raw_out + (long long)row * p.ld_raw + 4 * (lane + 64 * j)) = a;
      s += (a.x + a.y) + (a.z + a.w);
    }
  }
  s = wave_sum(s);
  const float mean = s / (float)p.N;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    if (lane + 64 * j < n4) {
      const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      q = fmaf(a, a, q); q = fmaf(b, b, q); q = fmaf(c, c, q); q = fmaf(d, d, q);
    }
  }
  q = wave_sum(q);
  const float rstd = 1.0f / sqrtf(q / (float)p.N + p.eps);
#pragma unroll
  for (int j = 0; j < JN; ++j) {
    if (lane + 64 * j < n4) {
      float4 y;
      y.x = fmaf((v[j].x - mean) * rstd, ga[j].x, be[j].x); y.y = fmaf((v[j].y - mean) * rstd, ga[j].y, be[j].y);
      y.z = fmaf((v[j].z - mean) * rstd, ga[j].z, be[j].z); y.w = fmaf((v[j].w - mean) * rstd, ga[j].w, be[j].w);
      *reinterpret_cast<float4*>(p.norm_out + (long long)row * p.ld_norm + 4 * (lane + 64 * j)) = y;
    }
  }
}

// ---- plain mode: elementwise reduce + epilogue -> raw_out (one float4 per thread)
__global__ __launch_bounds__(256) void post_plain_kernel(const ldmk_post_args p) {
  const int n4 = p.N >> 2;
  const long long total = (long long)p.M * n4;
  const long long i = min((long long)blockIdx.x * 256 + threadIdx.x, total - 1);
  const int row = (int)(i / n4), c = (int)(i - (long long)row * n4) * 4;
  const float* base = p.src + (long long)row * p.N + c;
  float4 a = f4_ld(base);
  const float4 eb = p.bias ? f4_ld(p.bias + c) : F4_ZERO;
  const float4 ev = p.batch_vec ? f4_ld(p.batch_vec + (long long)(row / p.rows_per_sample) * p.batch_vec_ld + c) : F4_ZERO;
  const float4 er = p.residual ? f4_ld(p.residual + (long long)row * p.ldr + c) : F4_ZERO;
  a = slab_rest4(a, base, p.nslab, p.slab_stride);
  a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
  if (p.bias) a = f4_add(a, eb);
  if (p.batch_vec) a = f4_add(a, ev);
  if (p.residual) a = f4_add(a, er);
  if ((long long)blockIdx.x * 256 + threadIdx.x < total) *reinterpret_cast<float4*>(p.raw_out + (long long)row * p.ld_raw + c) = a;
}

// ---- GEGLU mode: src columns are the packed (value | gate) 32-column pairs of ldmk_igemm's GEGLU layout;
// raw_out[M][N/2] = (value + bias_v) * gelu(gate + bias_g)      (attention.py:37-50)
__global__ __launch_bounds__(256) void post_geglu_kernel(const ldmk_post_args p) {
  const int half = p.N >> 1;
  const int h4 = half >> 2;
  const long long total = (long long)p.M * h4;
  const long long i = min((long long)blockIdx.x * 256 + threadIdx.x, total - 1);
  const int row = (int)(i / h4), c = (int)(i - (long long)row * h4) * 4;      // output column (4 of them: same 32-block)
  const int cv = ((c >> 5) << 6) + (c & 31), cg = cv + 32;                      // packed source columns
  const float* bv_ = p.src + (long long)row * p.N + cv;
  const float* bg_ = p.src + (long long)row * p.N + cg;
  float4 a = f4_ld(bv_), b = f4_ld(bg_);
  const float4 ea = p.bias ? f4_ld(p.bias + cv) : F4_ZERO, eg = p.bias ? f4_ld(p.bias + cg) : F4_ZERO;
  for (int k = 1; k < p.nslab; k += POST_SB) {
    float4 t[POST_SB], u_[POST_SB];
#pragma unroll
    for (int u = 0; u < POST_SB; ++u) {
      const long long so = (long long)min(k + u, p.nslab - 1) * p.slab_stride;
      t[u] = f4_ld(bv_ + so);
      u_[u] = f4_ld(bg_ + so);
    }
#pragma unroll
    for (int u = 0; u < POST_SB; ++u) {
      a = f4_add(a, k + u < p.nslab ? t[u] : F4_ZERO);
      b = f4_add(b, k + u < p.nslab ? u_[u] : F4_ZERO);
    }
  }
  a.x *= p.alpha; a.y *= p.alpha; a.z *= p.alpha; a.w *= p.alpha;
  b.x *= p.alpha; b.y *= p.alpha; b.z *= p.alpha; b.w *= p.alpha;
  if (p.bias) { a = f4_add(a, ea); b = f4_add(b, eg); }
  float4 y;
  y.x = a.x * gelu_erf_f(b.x); y.y = a.y * gelu_erf_f(b.y); y.z = a.z * gelu_erf_f(b.z); y.w = a.w * gelu_erf_f(b.w);
  if ((long long)blockIdx.x * 256 + threadIdx.x < total) *reinterpret_cast<float4*>(p.raw_out + (long long)row * p.ld_raw + c) = y;
}

}  // namespace ldmk

extern "C" long long ldmk_post_scratch_elems(const ldmk_post_args* args) {
  if (!args || args->rows_per_sample <= 0 || args->M <= 0) return -1;
  if (args->norm != LDMK_POST_GROUPNORM || args->rows_per_sample < LDMK_POST_GN_TILED_ROWS) return 0;
  return (long long)(args->M / args->rows_per_sample) * ((args->rows_per_sample + ldmk::GS_R - 1) / ldmk::GS_R) * 32 * 2;
}

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
    const long long g = ((long long)p.M * (p.N / 8) + 255) / 256;
    LDMK_REQUIRE(g < (1LL << 31), "ldmk_post: too many elements");
    hipLaunchKernelGGL(post_geglu_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
    return check_launch("ldmk_post(geglu)");
  }
  LDMK_REQUIRE(!p.raw_out || p.ld_raw >= p.N, "ldmk_post: raw_out row stride < N");
  if (p.norm == LDMK_POST_NONE) {
    LDMK_REQUIRE(p.raw_out != nullptr, "ldmk_post: nothing to write (raw_out and norm both off)");
    const long long g = ((long long)p.M * (p.N / 4) + 255) / 256;
    LDMK_REQUIRE(g < (1LL << 31), "ldmk_post: too many elements");
    hipLaunchKernelGGL(post_plain_kernel, dim3((unsigned)g), dim3(256), 0, st, p);
    return check_launch("ldmk_post(plain)");
  }
  LDMK_REQUIRE(p.gamma && p.beta && p.norm_out, "ldmk_post: norm needs gamma, beta and norm_out");
  if (p.norm == LDMK_POST_LAYERNORM) {
    LDMK_REQUIRE(p.N <= 2048 && !p.x1 && p.c1 == 0 && p.ld_norm >= p.N, "ldmk_post: LayerNorm rows of at most 2048 floats, no concat");
    const int jn = (p.N / 4 + 63) / 64;
    if (jn <= 1) hipLaunchKernelGGL(post_ln_kernel<1>, dim3(p.M), dim3(64), 0, st, p);
    else if (jn == 2) hipLaunchKernelGGL(post_ln_kernel<2>, dim3(p.M), dim3(64), 0, st, p);
    else if (jn == 3) hipLaunchKernelGGL(post_ln_kernel<3>, dim3(p.M), dim3(64), 0, st, p);
    else if (jn <= 5) hipLaunchKernelGGL(post_ln_kernel<5>, dim3(p.M), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(post_ln_kernel<8>, dim3(p.M), dim3(64), 0, st, p);
    return check_launch("ldmk_post(layernorm)");
  }
  LDMK_REQUIRE(p.norm == LDMK_POST_GROUPNORM, "ldmk_post: unknown norm %d", p.norm);
  const int C = p.N + p.c1;
  LDMK_REQUIRE((p.c1 == 0) == (p.x1 == nullptr) && p.c1 >= 0, "ldmk_post: x1 / c1 mismatch");
  LDMK_REQUIRE(p.groups > 0 && C % p.groups == 0 && p.ld_norm >= C && C / p.groups <= 64, "ldmk_post: C=%d groups=%d ld_norm=%d "
               "(at most 64 channels per group)", C, p.groups, p.ld_norm);
  if (p.rows_per_sample >= LDMK_POST_GN_TILED_ROWS) {      // large images: row-tiled statistics + apply launches
    const int chunks = (p.rows_per_sample + GS_R - 1) / GS_R;
    const int ns = p.M / p.rows_per_sample;
    const long long need = (long long)ns * chunks * 32 * 2;
    LDMK_REQUIRE(p.groups == 32 && p.c1 % 4 == 0 && C * GS_R * 4 <= 64 * 1024, "ldmk_post: the row-tiled GroupNorm takes 32 groups, "
                 "c1 %% 4 == 0 and at most %d channels (C=%d)", 64 * 1024 / (4 * GS_R), C);
    LDMK_REQUIRE_MEM(p.gn_scratch && p.gn_scratch_elems >= need, "ldmk_post: GroupNorm over %d rows per sample needs gn_scratch of %lld "
                     "floats (ldmk_post_scratch_elems), %lld given", p.rows_per_sample, need, p.gn_scratch ? p.gn_scratch_elems : 0LL);
    const bool plain_src = p.nslab == 1 && !p.bias && !p.batch_vec && !p.residual && p.alpha == 1.0f;
    LDMK_REQUIRE(p.raw_out || plain_src, "ldmk_post: a pending source needs raw_out (the apply launch reads the finished values)");
    if (plain_src) p.raw_out = nullptr;                       // nothing to store: the apply launch reads src itself
    hipLaunchKernelGGL(post_gnstat_kernel, dim3(chunks, ns), dim3(256), (size_t)GS_R * C * sizeof(float), st, p, p.gn_scratch);
    hipLaunchKernelGGL(post_gnapply_kernel, dim3(chunks, ns), dim3(256), 0, st, p, p.gn_scratch, chunks);
    return check_launch("ldmk_post(groupnorm, row-tiled)");
  }
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
