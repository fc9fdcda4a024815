// Backward / optimizer kernels of the UNet training step (ddpm.py:1014-1047 p_losses, :1363-1385
// AdamW, ema.py:25-44).  Everything here is HBM-bound elementwise or reduction work on NHWC fp32 tensors; the
// matrix products of the backward pass run in igemm.hip (data gradients) and wgrad.hip (weight gradients).
// All reductions use fixed partial layouts and fixed summation orders: gradients are bitwise reproducible.
#include "ldmk_common.h"

namespace ldmk {

constexpr int GB_PIX = 64;   // pixels per GroupNorm-backward partial record

__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float z) {
  const float s = sigmoid_f(z);
  return s * (1.0f + z * (1.0f - s));
}

// ---- GroupNorm group statistics (mean, rstd) from the forward's partial records ------------------------------
__global__ __launch_bounds__(256) void gn_group_stats_kernel(const float* __restrict__ pa, int c0,
                                                             const float* __restrict__ pb, int c1, int hw, int chunks,
                                                             int groups, float eps, float* __restrict__ mr) {
  // one workgroup per (sample, group), same reduction as gn_finalize_kernel (norms.hip)
  __shared__ double red[2][4];
  const int C = c0 + c1, g = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cpg = C / groups;
  const int items = cpg * chunks;
  double sum = 0.0, sumsq = 0.0;
  for (int i = threadIdx.x; i < items; i += 256) {
    const int ch = i / cpg, cc = i - ch * cpg, c = g * cpg + cc;
    const float* d = c < c0 ? pa + (((long long)n * chunks + ch) * c0 + c) * 3
                            : pb + (((long long)n * chunks + ch) * c1 + (c - c0)) * 3;
    const int cnt = min(hw - ch * 32, 32);
    const double sh = d[0], s = d[1], ss = d[2];
    sum += s + cnt * sh;
    sumsq += ss + 2.0 * sh * s + cnt * sh * sh;
  }
  sum = wave_sum_d(sum);
  sumsq = wave_sum_d(sumsq);
  if (lane == 0) { red[0][wave] = sum; red[1][wave] = sumsq; }
  __syncthreads();
  if (threadIdx.x == 0) {
    sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    sumsq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const double cnt = (double)cpg * hw, mean = sum / cnt;
    double var = sumsq / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    mr[((long long)n * groups + g) * 2] = (float)mean;
    mr[((long long)n * groups + g) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// ---- GroupNorm(+SiLU) backward, pass 1: per (sample, 64-pixel chunk, channel) sums of dz and dz*xhat ----------
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(const float* __restrict__ x0, int c0,
                                                             const float* __restrict__ x1, int c1,
                                                             const float* __restrict__ dy, const float* __restrict__ coef,
                                                             const float* __restrict__ mr, int hw, int groups, int silu,
                                                             float* __restrict__ partial) {
  // 64 channels x 4 pixel lanes per workgroup: four rows of the chunk are in flight per channel, folded through LDS
  __shared__ float red[4][64][2];
  const int C = c0 + c1, cpg = C / groups;
  const int cl = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, chunk = blockIdx.y, n = blockIdx.z;
  const int chunks = gridDim.y;
  float a = 0.f, b = 0.f;
  if (c < C) {
    const float sc = coef[((long long)n * 2) * C + c], sh = coef[((long long)n * 2 + 1) * C + c];
    const int g = c / cpg;
    const float mean = mr[((long long)n * groups + g) * 2], rstd = mr[((long long)n * groups + g) * 2 + 1];
    const float* xs = c < c0 ? x0 + c : x1 + (c - c0);
    const int cs = c < c0 ? c0 : c1;
    const int p0 = chunk * GB_PIX, p1 = min(hw, p0 + GB_PIX);
    for (int p = p0 + pl; p < p1; p += 4) {
      const long long row = (long long)n * hw + p;
      const float xv = xs[row * cs];
      float dz = dy[row * C + c];
      if (silu) dz *= dsilu_f(fmaf(xv, sc, sh));
      a += dz;
      b = fmaf(dz, (xv - mean) * rstd, b);
    }
  }
  red[pl][cl][0] = a;
  red[pl][cl][1] = b;
  __syncthreads();
  if (pl == 0 && c < C) {
    float* d = partial + (((long long)n * chunks + chunk) * C + c) * 2;
    d[0] = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
    d[1] = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
  }
}

// pass 2: per (sample, channel) totals over the chunks; per (sample, group) means of dxhat and dxhat*xhat
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, int C, int hw, int chunks,
                                                              int groups, const float* __restrict__ gamma,
                                                              float* __restrict__ tot, float* __restrict__ gstat) {
  const int n = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cpg = C / groups;
  for (int g = blockIdx.x * 4 + wave; g < groups; g += gridDim.x * 4) {
    float s1 = 0.f, s2 = 0.f;
    for (int cc = lane; cc < cpg; cc += 64) {
      const int c = g * cpg + cc;
      float a = 0.f, b = 0.f;
      for (int ch = 0; ch < chunks; ++ch) {
        const float* d = partial + (((long long)n * chunks + ch) * C + c) * 2;
        a += d[0]; b += d[1];
      }
      tot[((long long)n * C + c) * 2] = a;
      tot[((long long)n * C + c) * 2 + 1] = b;
      s1 = fmaf(gamma[c], a, s1);
      s2 = fmaf(gamma[c], b, s2);
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
      const float inv = 1.0f / ((float)cpg * (float)hw);
      gstat[((long long)n * groups + g) * 2] = s1 * inv;
      gstat[((long long)n * groups + g) * 2 + 1] = s2 * inv;
    }
  }
}

// dgamma[c] (+)= sum_n tot[n][c].b ; dbeta[c] (+)= sum_n tot[n][c].a
__global__ void gn_bwd_params_kernel(const float* __restrict__ tot, int n, int C, float* __restrict__ dgamma,
                                     float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float a = 0.f, b = 0.f;
  for (int i = 0; i < n; ++i) {
    a += tot[((long long)i * C + c) * 2];
    b += tot[((long long)i * C + c) * 2 + 1];
  }
  dgamma[c] = accumulate ? dgamma[c] + b : b;
  dbeta[c] = accumulate ? dbeta[c] + a : a;
}

// pass 3: dx = scale*dz - rstd*(m1 + xhat*m2), split back onto the two sources of the channel concat
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ x0, int c0, const float* __restrict__ x1,
                                                           int c1, const float* __restrict__ dy,
                                                           const float* __restrict__ coef, const float* __restrict__ mr,
                                                           const float* __restrict__ gstat, int hw, int groups, int silu,
                                                           float* __restrict__ dx0, int acc0, float* __restrict__ dx1,
                                                           int acc1, long long total) {
  const int C = c0 + c1, cpg = C / groups;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / C;
    const int c = (int)(i - row * C), n = (int)(row / hw), g = c / cpg;
    const float sc = coef[((long long)n * 2) * C + c], sh = coef[((long long)n * 2 + 1) * C + c];
    const float mean = mr[((long long)n * groups + g) * 2], rstd = mr[((long long)n * groups + g) * 2 + 1];
    const float m1 = gstat[((long long)n * groups + g) * 2], m2 = gstat[((long long)n * groups + g) * 2 + 1];
    const bool first = c < c0;
    const long long o = first ? row * c0 + c : row * c1 + (c - c0);
    const float xv = first ? x0[o] : x1[o];
    float dz = dy[i];
    if (silu) dz *= dsilu_f(fmaf(xv, sc, sh));
    const float v = sc * dz - rstd * fmaf((xv - mean) * rstd, m2, m1);
    float* d = first ? dx0 + o : dx1 + o;
    const int acc = first ? acc0 : acc1;
    *d = acc ? *d + v : v;
  }
}

// ---- LayerNorm: materialised forward and backward ---------------------------------------------------------
__global__ __launch_bounds__(256) void ln_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ y, int C, long long total4) {
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const float mu = stats[2 * row], rs = stats[2 * row + 1];
    const float4 v = *reinterpret_cast<const float4*>(x + row * C + c);
    const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
    float4 o;
    o.x = (v.x - mu) * rs * g.x + b.x; o.y = (v.y - mu) * rs * g.y + b.y;
    o.z = (v.z - mu) * rs * g.z + b.z; o.w = (v.w - mu) * rs * g.w + b.w;
    *reinterpret_cast<float4*>(y + row * C + c) = o;
  }
}

// LayerNorm backward.  One half-wave per row (32 lanes x float4 = 512 B per load instruction, C <= 1024, C % 4 == 0),
// a strip of LN_ROWS_PER_BLOCK rows per workgroup; per-workgroup column partials of dgamma / dbeta.
constexpr int LN_ROWS_PER_BLOCK = 64;
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ stats, const float* __restrict__ gamma,
                                                     float* __restrict__ dx, int accumulate, int rows, int C,
                                                     float* __restrict__ partial) {
  __shared__ float red[4][1024 * 2];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, half = lane >> 5;
  const int r0 = blockIdx.x * LN_ROWS_PER_BLOCK, r1 = min(rows, r0 + LN_ROWS_PER_BLOCK);
  const int c4n = C >> 2;
  float4 dg[8], db[8], gam[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = dg[i];
    const int c4 = l31 + 32 * i;
    gam[i] = c4 < c4n ? reinterpret_cast<const float4*>(gamma)[c4] : dg[i];
  }
  const float invC = 1.0f / (float)C;
  for (int row = r0 + wave * 2 + half; row < r1; row += 8) {
    const float mu = stats[2 * (long long)row], rs = stats[2 * (long long)row + 1];
    const float4* xr = reinterpret_cast<const float4*>(x + (long long)row * C);
    const float4* dyr = reinterpret_cast<const float4*>(dy + (long long)row * C);
    float4 xh[8], gv[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = l31 + 32 * i;
      xh[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      gv[i] = xh[i];
      if (c4 < c4n) {
        const float4 d = dyr[c4], xv = xr[c4];
        xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
        gv[i] = make_float4(d.x * gam[i].x, d.y * gam[i].y, d.z * gam[i].z, d.w * gam[i].w);
        dg[i].x = fmaf(d.x, xh[i].x, dg[i].x); dg[i].y = fmaf(d.y, xh[i].y, dg[i].y);
        dg[i].z = fmaf(d.z, xh[i].z, dg[i].z); dg[i].w = fmaf(d.w, xh[i].w, dg[i].w);
        db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
        s1 += (gv[i].x + gv[i].y) + (gv[i].z + gv[i].w);
        s2 = fmaf(gv[i].x, xh[i].x, s2); s2 = fmaf(gv[i].y, xh[i].y, s2);
        s2 = fmaf(gv[i].z, xh[i].z, s2); s2 = fmaf(gv[i].w, xh[i].w, s2);
      }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {        // stays inside the 32-lane half
      s1 += __shfl_xor(s1, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    s1 *= invC;
    s2 *= invC;
    float4* dxr = reinterpret_cast<float4*>(dx + (long long)row * C);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c4 = l31 + 32 * i;
      if (c4 < c4n) {
        float4 v = make_float4(rs * (gv[i].x - s1 - xh[i].x * s2), rs * (gv[i].y - s1 - xh[i].y * s2),
                               rs * (gv[i].z - s1 - xh[i].z * s2), rs * (gv[i].w - s1 - xh[i].w * s2));
        if (accumulate) {
          const float4 o = dxr[c4];
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        dxr[c4] = v;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {               // fold the two half-waves, then the four waves through LDS
    dg[i].x += __shfl_xor(dg[i].x, 32, 64); dg[i].y += __shfl_xor(dg[i].y, 32, 64);
    dg[i].z += __shfl_xor(dg[i].z, 32, 64); dg[i].w += __shfl_xor(dg[i].w, 32, 64);
    db[i].x += __shfl_xor(db[i].x, 32, 64); db[i].y += __shfl_xor(db[i].y, 32, 64);
    db[i].z += __shfl_xor(db[i].z, 32, 64); db[i].w += __shfl_xor(db[i].w, 32, 64);
    const int c4 = l31 + 32 * i;
    if (half == 0 && c4 < c4n) {
      float* d = &red[wave][8 * c4];
      d[0] = dg[i].x; d[1] = db[i].x; d[2] = dg[i].y; d[3] = db[i].y;
      d[4] = dg[i].z; d[5] = db[i].z; d[6] = dg[i].w; d[7] = db[i].w;
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) { a += red[w][2 * c]; b += red[w][2 * c + 1]; }
    partial[((long long)blockIdx.x * C + c) * 2] = a;
    partial[((long long)blockIdx.x * C + c) * 2 + 1] = b;
  }
}

// dgamma/dbeta = column sums of the per-workgroup partials: 64 columns x 4 row lanes per workgroup, LDS fold (fixed order)
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(const float* __restrict__ partial, int blocks, int C,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
  __shared__ float red[4][64][2];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
  float a = 0.f, b = 0.f;
  if (c < C)
    for (int i = rl; i < blocks; i += 4) {
      a += partial[((long long)i * C + c) * 2];
      b += partial[((long long)i * C + c) * 2 + 1];
    }
  red[rl][cl][0] = a;
  red[rl][cl][1] = b;
  __syncthreads();
  if (rl == 0 && c < C) {
    a = (red[0][cl][0] + red[1][cl][0]) + (red[2][cl][0] + red[3][cl][0]);
    b = (red[0][cl][1] + red[1][cl][1]) + (red[2][cl][1] + red[3][cl][1]);
    dgamma[c] = accumulate ? dgamma[c] + a : a;
    dbeta[c] = accumulate ? dbeta[c] + b : b;
  }
}

// ---- GEGLU (attention.py:37-45): pre = [value | gate], f = value * gelu(gate), exact erf GELU ---------------
__device__ __forceinline__ float gelu_f(float g) { return 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float g) {
  return 0.5f * (1.0f + erff(g * 0.70710678118654752440f)) + g * 0.39894228040143267794f * __expf(-0.5f * g * g);
}
__global__ void geglu_fwd_kernel(const float* __restrict__ pre, float* __restrict__ f, int inner, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / inner;
    const int c = (int)(i - row * inner);
    const float* p = pre + row * 2 * inner;
    f[i] = p[c] * gelu_f(p[inner + c]);
  }
}
__global__ void geglu_bwd_kernel(const float* __restrict__ pre, const float* __restrict__ df, float* __restrict__ dpre,
                                 int inner, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / inner;
    const int c = (int)(i - row * inner);
    const float* p = pre + row * 2 * inner;
    const float v = p[c], g = p[inner + c], d = df[i];
    dpre[row * 2 * inner + c] = d * gelu_f(g);
    dpre[row * 2 * inner + inner + c] = d * v * dgelu_f(g);
  }
}

// ---- softmax backward on rows: ds = p * (dp - sum_j dp_j p_j) * scale (in place over dp) -----------------------
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ p, float* __restrict__ dp,
                                                               int cols, float scale) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const float* pr = p + row * cols;
  float* dr = dp + row * cols;
  float s = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) s = fmaf(pr[c], dr[c], s);
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  const float tot = (red[0] + red[1]) + (red[2] + red[3]);
  for (int c = threadIdx.x; c < cols; c += 256) dr[c] = pr[c] * (dr[c] - tot) * scale;
}

// ---- column sums over row groups: out[g][n] (+)= sum_{r in group g} x[r][n] (bias and per-sample-vector grads) --
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, int ldx, int rows_per_group,
                                                             int N, int splits, float* __restrict__ partial) {
  __shared__ float red[4][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int g = blockIdx.y, sp = blockIdx.z;
  const int per = (rows_per_group + splits - 1) / splits;
  const int r0 = sp * per, r1 = min(rows_per_group, r0 + per);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) s += x[((long long)g * rows_per_group + r) * ldx + c];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && c < N)
    partial[((long long)g * splits + sp) * N + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int N, int splits, float* __restrict__ out, int ldo,
                                    int accumulate, int groups) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)groups * N) return;
  const int g = (int)(i / N), c = (int)(i - (long long)g * N);
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += partial[((long long)g * splits + k) * N + c];
  float* d = out + (long long)g * ldo + c;
  *d = accumulate ? *d + s : s;
}

// ---- nearest-x2 upsample backward: dx[n][y][x][c] (+)= sum of the 2x2 block of du ---------------------------
__global__ void sumpool2_kernel(const float* __restrict__ du, float* __restrict__ dx, int h, int w, int C, int accumulate,
                                long long total4) {
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long long pix = i / c4n;
    const int xx = (int)(pix % w);
    const long long t = pix / w;
    const int yy = (int)(t % h);
    const long long n = t / h;
    const float* s = du + (((n * 2 * h + 2 * yy) * 2 * w) + 2 * xx) * C + c;
    const float4 a = *reinterpret_cast<const float4*>(s), b = *reinterpret_cast<const float4*>(s + C);
    const float4 d = *reinterpret_cast<const float4*>(s + (long long)2 * w * C), e = *reinterpret_cast<const float4*>(s + (long long)2 * w * C + C);
    float4 o = make_float4((a.x + b.x) + (d.x + e.x), (a.y + b.y) + (d.y + e.y), (a.z + b.z) + (d.z + e.z), (a.w + b.w) + (d.w + e.w));
    float* dst = dx + pix * C + c;
    if (accumulate) { o.x += dst[0]; o.y += dst[1]; o.z += dst[2]; o.w += dst[3]; }
    *reinterpret_cast<float4*>(dst) = o;
  }
}

// ---- small elementwise pieces ---------------------------------------------------------------------------
__global__ void silu_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = silu_f(x[i]);
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dx[i] = dy[i] * dsilu_f(x[i]);
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, float a, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = fmaf(a, x[i], y[i]);
}

// q_sample (ddpm.py:1009-1012): x_t = sqrt_ac[t_b]*x0 + sqrt_1mac[t_b]*noise
__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const long long* __restrict__ t,
                                const float* __restrict__ sqrt_ac, const float* __restrict__ sqrt_1mac, float* __restrict__ xt,
                                int per, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long tb = t[i / per];
    xt[i] = sqrt_ac[tb] * x0[i] + sqrt_1mac[tb] * noise[i];
  }
}

// mean-squared-error loss (ddpm.py:324-334 get_loss 'l2' + .mean): dpred = 2*(pred-target)/n, loss partials per workgroup
__global__ __launch_bounds__(256) void mse_grad_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                       float* __restrict__ dpred, long long n, float gscale,
                                                       double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0.0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = pred[i] - target[i];
    dpred[i] = d * gscale;
    s += (double)d * d;
  }
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
__global__ void mse_final_kernel(const double* __restrict__ partial, int blocks, double inv_n, float* __restrict__ loss) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < blocks; ++i) s += partial[i];
    *loss = (float)(s * inv_n);
  }
}

// AdamW (torch.optim.AdamW semantics, ddpm.py:1363-1385): decoupled weight decay, bias-corrected moments
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             long long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    float pi = p[i] * (1.0f - lr * wd);
    const float mi = b1 * m[i] + (1.0f - b1) * gi;
    const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi;
  }
}
// LitEma.forward (ema.py:25-44): shadow -= (1-decay) * (shadow - param)
__global__ void ema_kernel(float* __restrict__ shadow, const float* __restrict__ p, long long n, float one_minus_decay) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
    shadow[i] -= one_minus_decay * (shadow[i] - p[i]);
}

// packed forward conv weights [Cin/32][9][32][Cout] -> packed data-gradient weights [Cout/32][9][32][Cin] with the
// taps mirrored: dX = conv3x3(dY, W') (stride-1 'same' convolution) -- one 32x32 LDS transpose per (chunk pair, tap)
__global__ __launch_bounds__(256) void pack_dgrad3x3_kernel(const float* __restrict__ wf, float* __restrict__ wd, int cin, int cout) {
  __shared__ float tile[32][33];
  const int ci_chunk = blockIdx.x, co_chunk = blockIdx.y, tap = blockIdx.z;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 8 rows per pass
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    tile[r][tx] = wf[(((long long)ci_chunk * 9 + tap) * 32 + r) * cout + co_chunk * 32 + tx];     // [ci32=r][co32=tx]
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8)
    wd[(((long long)co_chunk * 9 + (8 - tap)) * 32 + r) * cin + ci_chunk * 32 + tx] = tile[tx][r];  // [co32=r][ci32=tx]
}

// [n][T][parts][H][32] <-> [parts][n*H][T][32]: per-head contiguous matrices for the batched attention-backward GEMMs
__global__ void head_permute_kernel(const float* __restrict__ src, float* __restrict__ dst, int n, int T, int parts, int H,
                                    int to_heads, long long total4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i & 7);
    long long r = i >> 3;                    // index over (b, t, p, h) in token-major order
    const int h = (int)(r % H); r /= H;
    const int p = (int)(r % parts); r /= parts;
    const int t = (int)(r % T);
    const int b = (int)(r / T);
    const long long tok = i * 4;                                                              // token-major offset
    const long long hd = ((((long long)p * n + b) * H + h) * T + t) * 32 + d4 * 4;            // head-major offset
    if (to_heads) *reinterpret_cast<float4*>(dst + hd) = *reinterpret_cast<const float4*>(src + tok);
    else *reinterpret_cast<float4*>(dst + tok) = *reinterpret_cast<const float4*>(src + hd);
  }
}


// ---- Conv1DTemporalAttention backward (talking_face/ldm/modules/encoders/modules.py:76-113) -----------------------
// One workgroup per sample: the forward is recomputed into LDS (all five activations kept), then softmax / Linear(T,T) /
// the five Conv1d(k=3)+LeakyReLU(0.02) layers are walked back.  Parameter gradients are written per sample
// ([n][AAB_TOTAL(T, dim)] floats, packed [3][cin][cout] like the forward weights) and summed over the batch by the
// caller with ldmk_colsum; the audio features themselves need no gradient (wav2vec2 is frozen).
__device__ __forceinline__ float leaky(float v) { return v > 0.f ? v : 0.02f * v; }

__global__ __launch_bounds__(256) void audio_attention_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dout,
                                                                  int T, int dim, const float* const* __restrict__ w,
                                                                  const float* const* __restrict__ bias,
                                                                  const float* __restrict__ wl, const float* __restrict__ bl,
                                                                  float* __restrict__ grads, long long gstride) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float red[4];
  const int chans[6] = {dim, 192, 64, 16, 4, 1};
  const int Tp = T + 2, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
  float* a[6];
  float* da[6];
  float* ptr = sm;
  for (int l = 0; l < 6; ++l) { a[l] = ptr; ptr += Tp * chans[l]; }
  da[0] = nullptr;
  for (int l = 1; l < 6; ++l) { da[l] = ptr; ptr += Tp * chans[l]; }
  float* att = ptr;            // [T]
  float* dlog = att + T;       // [T]
  const float* xb = x + (long long)b * T * dim;
  float* g = grads + (long long)b * gstride;
  // ---- forward recompute (padding rows 0 and T+1 are zero)
  for (int i = tid; i < Tp * dim; i += 256) {
    const int t = i / dim - 1, c = i - (t + 1) * dim;
    a[0][i] = (t >= 0 && t < T) ? xb[(long long)t * dim + c] : 0.f;
  }
  for (int l = 1; l < 6; ++l)
    for (int i = tid; i < Tp * chans[l]; i += 256) { a[l][i] = 0.f; da[l][i] = 0.f; }
  __syncthreads();
  for (int l = 0; l < 5; ++l) {
    const int cin = chans[l], cout = chans[l + 1];
    for (int o = tid; o < T * cout; o += 256) {
      const int t = o / cout, co = o - t * cout;
      float acc = bias[l][co];
      for (int k = 0; k < 3; ++k) {
        const float* src = a[l] + (t + k) * cin;
        const float* wk = w[l] + (long long)k * cin * cout + co;
        for (int ci = 0; ci < cin; ++ci) acc = fmaf(src[ci], wk[(long long)ci * cout], acc);
      }
      a[l + 1][(t + 1) * cout + co] = leaky(acc);
    }
    __syncthreads();
  }
  if (tid < T) {
    float v = bl[tid];
    for (int j = 0; j < T; ++j) v = fmaf(wl[tid * T + j], a[5][j + 1], v);
    att[tid] = v;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY, s_ = 0.f;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, att[t]);
    for (int t = 0; t < T; ++t) { att[t] = expf(att[t] - mx); s_ += att[t]; }
    for (int t = 0; t < T; ++t) att[t] /= s_;
  }
  __syncthreads();
  // ---- d att[t] = sum_c dout[c] x[t][c]
  const float* db_ = dout + (long long)b * dim;
  for (int t = 0; t < T; ++t) {
    float p = 0.f;
    for (int c = tid; c < dim; c += 256) p = fmaf(db_[c], a[0][(t + 1) * dim + c], p);
    p = wave_sum(p);
    if (lane == 0) red[wave] = p;
    __syncthreads();
    if (tid == 0) dlog[t] = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
  }
  if (tid == 0) {              // softmax backward in place: dlog <- att * (datt - sum att*datt)
    float s_ = 0.f;
    for (int t = 0; t < T; ++t) s_ = fmaf(att[t], dlog[t], s_);
    for (int t = 0; t < T; ++t) dlog[t] = att[t] * (dlog[t] - s_);
  }
  __syncthreads();
  // ---- gradient layout per sample: conv weights (packed [3][cin][cout]) l = 0..4, conv biases l = 0..4, Linear W, b
  long long off = 0;
  float* gw[5];
  for (int l = 0; l < 5; ++l) { gw[l] = g + off; off += 3LL * chans[l] * chans[l + 1]; }
  float* gb[5];
  for (int l = 0; l < 5; ++l) { gb[l] = g + off; off += chans[l + 1]; }
  float* gwl = g + off;
  float* gbl = gwl + T * T;
  for (int i = tid; i < T * T; i += 256) gwl[i] = dlog[i / T] * a[5][i % T + 1];
  if (tid < T) {
    gbl[tid] = dlog[tid];
    float v = 0.f;
    for (int t = 0; t < T; ++t) v = fmaf(wl[t * T + tid], dlog[t], v);
    da[5][tid + 1] = v;                                   // d a5[j], cout = 1
  }
  __syncthreads();
  for (int l = 4; l >= 0; --l) {
    const int cin = chans[l], cout = chans[l + 1];
    // dz = da * leaky'(z): the activation output has the sign of its input
    for (int i = tid; i < T * cout; i += 256) {
      const int idx = (i / cout + 1) * cout + (i % cout);
      da[l + 1][idx] *= a[l + 1][idx] > 0.f ? 1.f : 0.02f;
    }
    __syncthreads();
    for (int co = tid; co < cout; co += 256) {
      float v = 0.f;
      for (int t = 0; t < T; ++t) v += da[l + 1][(t + 1) * cout + co];
      gb[l][co] = v;
    }
    for (long long i = tid; i < 3LL * cin * cout; i += 256) {       // dW[k][ci][co] = sum_t a_l[t+k][ci] dz[t][co]
      const int co = (int)(i % cout);
      const long long r = i / cout;
      const int ci = (int)(r % cin), k = (int)(r / cin);
      float v = 0.f;
      for (int t = 0; t < T; ++t) v = fmaf(a[l][(t + k) * cin + ci], da[l + 1][(t + 1) * cout + co], v);
      gw[l][i] = v;
    }
    if (l > 0) {                                                    // d a_l[s][ci] = sum_k sum_co W[k][ci][co] dz[s-k][co]
      for (int i = tid; i < T * cin; i += 256) {
        const int t = i / cin, ci = i - t * cin, s_ = t + 1;         // padded row index of a_l
        float v = 0.f;
        for (int k = 0; k < 3; ++k) {
          const int tz = s_ - k;                                    // output time whose tap k read padded row s_
          if (tz < 0 || tz >= T) continue;
          const float* wk = w[l] + ((long long)k * cin + ci) * cout;
          const float* dz = da[l + 1] + (tz + 1) * cout;
          for (int co = 0; co < cout; ++co) v = fmaf(wk[co], dz[co], v);
        }
        da[l][s_ * cin + ci] = v;
      }
    }
    __syncthreads();
  }
}

static inline int grid_for(long long n) {
  long long g = (n + 255) / 256;
  return (int)(g > 8192 ? 8192 : (g < 1 ? 1 : g));
}

}  // namespace ldmk

using namespace ldmk;

extern "C" int ldmk_gn_bwd_chunks(int hw) { return (hw + GB_PIX - 1) / GB_PIX; }

extern "C" int ldmk_gn_group_stats(const float* partial0, int c0, const float* partial1, int c1, int n, int hw, int groups,
                                   float eps, float* mr, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(partial0 && mr && c0 > 0 && n > 0 && hw > 0 && groups > 0 && (c0 + c1) % groups == 0, "ldmk_gn_group_stats: bad args");
  LDMK_REQUIRE((c1 == 0) == (partial1 == nullptr), "ldmk_gn_group_stats: partial1/c1 mismatch");
  hipLaunchKernelGGL(gn_group_stats_kernel, dim3(groups, n), dim3(256), 0, (hipStream_t)stream, partial0, c0, partial1,
                     c1, hw, ldmk_gn_chunks(hw), groups, eps, mr);
  return check_launch("ldmk_gn_group_stats");
}

extern "C" int ldmk_gn_bwd(const float* x0, int c0, const float* x1, int c1, const float* dy, const float* coef,
                           const float* mr, const float* gamma, int n, int hw, int groups, int silu, float* dx0, int acc0,
                           float* dx1, int acc1, float* dgamma, float* dbeta, int acc_params, float* scratch, void* stream) {
  LDMK_ENTER();
  const int C = c0 + c1;
  LDMK_REQUIRE(x0 && dy && coef && mr && gamma && dx0 && dgamma && dbeta && scratch, "ldmk_gn_bwd: null buffer");
  LDMK_REQUIRE(c0 > 0 && n > 0 && hw > 0 && groups > 0 && C % groups == 0, "ldmk_gn_bwd: bad shape");
  LDMK_REQUIRE((c1 == 0) == (x1 == nullptr) && (c1 == 0) == (dx1 == nullptr), "ldmk_gn_bwd: second source mismatch");
  const int chunks = ldmk_gn_bwd_chunks(hw);
  float* partial = scratch;                                  // [n][chunks][C][2]
  float* tot = partial + (long long)n * chunks * C * 2;      // [n][C][2]
  float* gstat = tot + (long long)n * C * 2;                 // [n][groups][2]
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3((C + 63) / 64, chunks, n), dim3(256), 0, st, x0, c0, x1, c1, dy, coef, mr, hw,
                     groups, silu, partial);
  hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((groups + 3) / 4, n), dim3(256), 0, st, partial, C, hw, chunks, groups, gamma,
                     tot, gstat);
  hipLaunchKernelGGL(gn_bwd_params_kernel, dim3((C + 255) / 256), dim3(256), 0, st, tot, n, C, dgamma, dbeta, acc_params);
  const long long total = (long long)n * hw * C;
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, st, x0, c0, x1, c1, dy, coef, mr, gstat, hw,
                     groups, silu, dx0, acc0, dx1, acc1, total);
  return check_launch("ldmk_gn_bwd");
}
extern "C" long long ldmk_gn_bwd_scratch_elems(int n, int hw, int c, int groups) {
  return (long long)n * ldmk_gn_bwd_chunks(hw) * c * 2 + (long long)n * c * 2 + (long long)n * groups * 2;
}

extern "C" int ldmk_ln_apply(const float* x, const float* stats, const float* gamma, const float* beta, float* y, int rows,
                             int c, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && stats && gamma && beta && y && rows > 0 && c > 0 && c % 4 == 0, "ldmk_ln_apply: bad args (C%%4==0)");
  const long long total4 = (long long)rows * (c / 4);
  hipLaunchKernelGGL(ln_apply_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, stats, gamma, beta, y, c, total4);
  return check_launch("ldmk_ln_apply");
}
extern "C" int ldmk_ln_bwd_blocks(int rows) { return (rows + LN_ROWS_PER_BLOCK - 1) / LN_ROWS_PER_BLOCK; }
extern "C" int ldmk_ln_bwd(const float* dy, const float* x, const float* stats, const float* gamma, float* dx, int acc_dx,
                           int rows, int c, float* dgamma, float* dbeta, int acc_params, float* scratch, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(dy && x && stats && gamma && dx && dgamma && dbeta && scratch, "ldmk_ln_bwd: null buffer");
  LDMK_REQUIRE(rows > 0 && c > 0 && c <= 1024 && c % 4 == 0, "ldmk_ln_bwd: bad shape (C<=1024, C%%4==0)");
  const int blocks = ldmk_ln_bwd_blocks(rows);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(blocks), dim3(256), 0, st, dy, x, stats, gamma, dx, acc_dx, rows, c, scratch);
  hipLaunchKernelGGL(ln_bwd_params_kernel, dim3((c + 63) / 64), dim3(256), 0, st, scratch, blocks, c, dgamma, dbeta, acc_params);
  return check_launch("ldmk_ln_bwd");
}

extern "C" int ldmk_geglu_fwd(const float* pre, float* f, long long rows, int inner, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(pre && f && rows > 0 && inner > 0, "ldmk_geglu_fwd: bad args");
  const long long total = rows * inner;
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, pre, f, inner, total);
  return check_launch("ldmk_geglu_fwd");
}
extern "C" int ldmk_geglu_bwd(const float* pre, const float* df, float* dpre, long long rows, int inner, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(pre && df && dpre && rows > 0 && inner > 0, "ldmk_geglu_bwd: bad args");
  const long long total = rows * inner;
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, pre, df, dpre, inner, total);
  return check_launch("ldmk_geglu_bwd");
}

extern "C" int ldmk_softmax_bwd_rows(const float* p, float* dp, long long rows, int cols, float scale, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(p && dp && rows > 0 && rows <= 0x7fffffffLL && cols > 0, "ldmk_softmax_bwd_rows: bad args");
  hipLaunchKernelGGL(softmax_bwd_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, p, dp, cols, scale);
  return check_launch("ldmk_softmax_bwd_rows");
}

extern "C" int ldmk_colsum_splits(int rows_per_group) {
  int s = rows_per_group / 256;
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}
extern "C" int ldmk_colsum(const float* x, int ldx, int rows_per_group, int groups, int n, float* out, int ldo, int accumulate,
                           float* scratch, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && out && scratch && rows_per_group > 0 && groups > 0 && groups <= 65535 && n > 0, "ldmk_colsum: bad args");
  const int splits = ldmk_colsum_splits(rows_per_group);
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3((n + 63) / 64, groups, splits), dim3(256), 0, st, x, ldx, rows_per_group, n,
                     splits, scratch);
  const long long tot = (long long)groups * n;
  hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, scratch, n, splits, out, ldo,
                     accumulate, groups);
  return check_launch("ldmk_colsum");
}

extern "C" int ldmk_sumpool2(const float* du, float* dx, int n, int h, int w, int c, int accumulate, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(du && dx && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "ldmk_sumpool2: bad args (C%%4==0)");
  const long long total4 = (long long)n * h * w * (c / 4);
  hipLaunchKernelGGL(sumpool2_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, du, dx, h, w, c, accumulate, total4);
  return check_launch("ldmk_sumpool2");
}

extern "C" int ldmk_silu(const float* x, float* y, long long n, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && y && n > 0, "ldmk_silu: bad args");
  hipLaunchKernelGGL(silu_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n);
  return check_launch("ldmk_silu");
}
extern "C" int ldmk_silu_bwd(const float* x, const float* dy, float* dx, long long n, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && dy && dx && n > 0, "ldmk_silu_bwd: bad args");
  hipLaunchKernelGGL(silu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
  return check_launch("ldmk_silu_bwd");
}
extern "C" int ldmk_axpy(float* y, const float* x, float a, long long n, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && y && n > 0, "ldmk_axpy: bad args");
  hipLaunchKernelGGL(axpy_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, x, a, n);
  return check_launch("ldmk_axpy");
}

extern "C" int ldmk_q_sample(const float* x0, const float* noise, const long long* t, const float* sqrt_ac,
                             const float* sqrt_1mac, float* xt, int n, int per, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x0 && noise && t && sqrt_ac && sqrt_1mac && xt && n > 0 && per > 0, "ldmk_q_sample: bad args");
  const long long total = (long long)n * per;
  hipLaunchKernelGGL(q_sample_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x0, noise, t, sqrt_ac, sqrt_1mac,
                     xt, per, total);
  return check_launch("ldmk_q_sample");
}

extern "C" int ldmk_mse_grad(const float* pred, const float* target, float* dpred, long long n, long long denom, float* loss,
                             double* scratch, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(pred && target && dpred && loss && scratch && n > 0 && denom >= 0, "ldmk_mse_grad: bad args");
  if (denom == 0) denom = n;                                 // channel-padded tensors: mean over the real elements only
  const int blocks = 256;                                    // scratch: 256 doubles
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mse_grad_kernel, dim3(blocks), dim3(256), 0, st, pred, target, dpred, n, (float)(2.0 / (double)denom), scratch);
  hipLaunchKernelGGL(mse_final_kernel, dim3(1), dim3(64), 0, st, scratch, blocks, 1.0 / (double)denom, loss);
  return check_launch("ldmk_mse_grad");
}

extern "C" int ldmk_head_permute(const float* src, float* dst, int n, int tokens, int parts, int heads, int to_heads,
                                 void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(src && dst && n > 0 && tokens > 0 && parts > 0 && heads > 0, "ldmk_head_permute: bad args");
  const long long total4 = (long long)n * tokens * parts * heads * 8;
  hipLaunchKernelGGL(head_permute_kernel, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, src, dst, n, tokens, parts,
                     heads, to_heads, total4);
  return check_launch("ldmk_head_permute");
}

extern "C" long long ldmk_audio_attention_grad_elems(int T, int dim) {
  const long long ch[6] = {dim, 192, 64, 16, 4, 1};
  long long e = 0;
  for (int l = 0; l < 5; ++l) e += 3 * ch[l] * ch[l + 1] + ch[l + 1];
  return e + (long long)T * T + T;
}

extern "C" int ldmk_audio_attention_bwd(const float* x, const float* dout, int n, int T, int dim, const float* const* conv_w,
                                        const float* const* conv_b, const float* lin_w, const float* lin_b,
                                        float* grads_per_sample, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && dout && conv_w && conv_b && lin_w && lin_b && grads_per_sample && n > 0 && dim > 0, "ldmk_audio_attention_bwd: bad args");
  LDMK_REQUIRE(T >= 1, "ldmk_audio_attention_bwd: T");
  const size_t lds = ((size_t)(T + 2) * (dim + 2 * (192 + 64 + 16 + 4 + 1)) + 2 * T) * sizeof(float);
  const size_t lds_max = 160 * 1024 - 256;       // the kernel also has a few bytes of static LDS
  LDMK_REQUIRE(lds <= lds_max, "ldmk_audio_attention_bwd: window T=%d too large for LDS (%zu bytes)", T, lds);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(audio_attention_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    attr = true;
  }
  hipLaunchKernelGGL(audio_attention_bwd_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, x, dout, T, dim, conv_w, conv_b,
                     lin_w, lin_b, grads_per_sample, ldmk_audio_attention_grad_elems(T, dim));
  return check_launch("ldmk_audio_attention_bwd");
}

extern "C" int ldmk_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                          float eps, float weight_decay, int step, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(p && g && m && v && n > 0 && step >= 1, "ldmk_adamw: bad args");
  const float bc1 = 1.0f - powf(beta1, (float)step), bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2s);
  return check_launch("ldmk_adamw");
}
extern "C" int ldmk_ema(float* shadow, const float* p, long long n, float one_minus_decay, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(shadow && p && n > 0, "ldmk_ema: bad args");
  hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, shadow, p, n, one_minus_decay);
  return check_launch("ldmk_ema");
}

extern "C" int ldmk_pack_dgrad3x3(const float* w_fwd, float* w_dgrad, int cin, int cout, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(w_fwd && w_dgrad && cin > 0 && cout > 0 && cin % 32 == 0 && cout % 32 == 0, "ldmk_pack_dgrad3x3: channels %%32");
  LDMK_REQUIRE(cout / 32 <= 65535, "ldmk_pack_dgrad3x3: grid limit");
  hipLaunchKernelGGL(pack_dgrad3x3_kernel, dim3(cin / 32, cout / 32, 9), dim3(256), 0, (hipStream_t)stream, w_fwd, w_dgrad, cin, cout);
  return check_launch("ldmk_pack_dgrad3x3");
}
