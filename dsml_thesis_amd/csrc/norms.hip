// GroupNorm / LayerNorm statistics (HBM-bound; the normalisation itself is applied by the
// consumer while it stages its A operand, see igemm.hip).
#include "ldmk_common.h"

namespace ldmk {

constexpr int GN_PIX = 32;   // pixels per partial-sum chunk == the MFMA row-tile, so igemm epilogues can emit partials

// pass 1 (stand-alone form; igemm / its split-K reduce emit the same records from their epilogue):
// per-(sample, chunk, channel) shifted sum and sum of squares.  Thread <-> channel, so a wave reads
// 64 consecutive floats of one NHWC pixel row: fully coalesced.
__global__ __launch_bounds__(256) void gn_partial_kernel(const float* __restrict__ x, int C, int hw, int chunks,
                                                         float* __restrict__ partial) {
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * GN_PIX;
  const int p1 = min(hw, p0 + GN_PIX);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float* ptr = x + ((long long)n * hw + p0) * C + c;
    // shifted sums (shift = first value of the chunk) keep the fp32 cancellation error small
    const float shift = ptr[0];
    float s = 0.f, ss = 0.f;
    for (int p = p0; p < p1; ++p) {
      float v = *ptr - shift;
      s += v;
      ss = fmaf(v, v, ss);
      ptr += C;
    }
    float* d = partial + (((long long)n * chunks + chunk) * C + c) * 3;
    d[0] = shift; d[1] = s; d[2] = ss;
  }
}

// pass 2: one wave per (sample, group): combine the chunk partials of (the channel concat of) up to two
// tensors in double -- groups may straddle the concat seam -- and emit per-channel scale/shift planes
//   y = x*scale + shift  ==  (x-mean)*rstd*gamma + beta.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ pa, int c0,
                                                          const float* __restrict__ pb, int c1, int hw, int chunks,
                                                          int groups, float eps, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ coef) {
  // one workgroup per (sample, group): at 64x64 a group has 5 channels x 128 chunk records -- one wave per group
  // walked them in 10 dependent-latency steps on 128 workgroups (25 us per call, 3.5 % of the step); 256 threads
  // take at most 3 records each and every CU gets work
  __shared__ double red[2][4];
  const int C = c0 + c1;
  const int g = blockIdx.x, n = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cpg = C / groups;
  const int items = cpg * chunks;
  // gamma / beta of this thread's output channel are requested BEFORE the record loop: they do not depend on it, and a load
  // issued after the reduction would be a second dependent memory round trip in a kernel that is nothing but latency
  const int cmine = g * cpg + min((int)threadIdx.x, cpg - 1);
  const float gam = gamma[cmine], bet = beta[cmine];
  double sum = 0.0, sumsq = 0.0;
  for (int i = threadIdx.x; i < items; i += 256) {
    const int ch = i / cpg, cc = i - ch * cpg;
    const int c = g * cpg + cc;
    const float* d = c < c0 ? pa + (((long long)n * chunks + ch) * c0 + c) * 3
                            : pb + (((long long)n * chunks + ch) * c1 + (c - c0)) * 3;
    const int cnt = min(hw - ch * GN_PIX, GN_PIX);
    const double sh = d[0], s = d[1], ss = d[2];
    // sum x = s + cnt*sh ; sum x^2 = ss + 2*sh*s + cnt*sh^2
    sum += s + cnt * sh;
    sumsq += ss + 2.0 * sh * s + cnt * sh * sh;
  }
  sum = wave_sum_d(sum);
  sumsq = wave_sum_d(sumsq);
  if (lane == 0) { red[0][wave] = sum; red[1][wave] = sumsq; }
  __syncthreads();
  sum = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
  sumsq = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  const double cnt = (double)cpg * hw;
  const double mean = sum / cnt;
  double var = sumsq / cnt - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  for (int cc = threadIdx.x; cc < cpg; cc += 256) {
    const int c = g * cpg + cc;
    const bool first = cc == (int)threadIdx.x;                 // (cpg <= 256 in every model here: one channel per thread)
    const float sc = rstd * (first ? gam : gamma[c]);
    coef[((long long)n * 2) * C + c] = sc;
    coef[((long long)n * 2 + 1) * C + c] = fmaf(-meanf, sc, first ? bet : beta[c]);
  }
}

// LayerNorm statistics: one half-wave (32 lanes x float4 = 512 B per load instruction) per row, exact
// two-pass in registers (C <= 1024, C % 4 == 0); the generic scalar form handles other widths.
typedef __bf16 nbf16x4 __attribute__((ext_vector_type(4)));
// SPLIT: also write the row as the three bf16 images of its exact split x = hi + mid + lo (round to nearest even, the split
// igemm_kernel<BF = 3> makes while staging): split[img][row][ld], the a_split operand of LDMK_COMPUTE_BF16X3
// GUARD (flag != nullptr): rows whose |mean| exceeds `guard` standard deviations set *flag -- the consumer that folds the
// LayerNorm through its product (LDMK_TF_LAYERNORM_FOLDED) subtracts mean * colsum(W') from x W' in fp32 and loses
// ~|mean| / std ulps on such rows; the host reads the flag once per program and switches the model to the unfolded prologue.
template <bool SPLIT>
__global__ __launch_bounds__(256) void ln_stats_kernel(const float* __restrict__ x, int rows, int C, float eps,
                                                       float* __restrict__ stats, __bf16* __restrict__ split, int ld_split,
                                                       float guard, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63, l31 = lane & 31;
  const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
  const bool ok = row < rows;
  const float4* p = reinterpret_cast<const float4*>(x + (ok ? row : 0) * C);
  const int c4n = C >> 2;
  float4 v[8];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = l31 + 32 * i;
    v[i] = (ok && c4 < c4n) ? p[c4] : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);     // stays inside the 32-lane half
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c4 = l31 + 32 * i;
    if (c4 < c4n) {
      const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
      q = fmaf(a, a, q); q = fmaf(b, b, q); q = fmaf(c, c, q); q = fmaf(d, d, q);
    }
  }
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
  if (ok && l31 == 0) {
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
    if (flag && !(fabsf(mean) * rstd <= guard)) *flag = 1;       // (NaN statistics raise it too)
  }
  if constexpr (SPLIT) {
    if (ok) {
      const long long img = (long long)rows * ld_split;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int c4 = l31 + 32 * i;
        if (c4 < c4n) {
          const nbf16x4 h = {(__bf16)v[i].x, (__bf16)v[i].y, (__bf16)v[i].z, (__bf16)v[i].w};
          const float r0 = v[i].x - (float)h[0], r1 = v[i].y - (float)h[1], r2 = v[i].z - (float)h[2], r3 = v[i].w - (float)h[3];
          const nbf16x4 m = {(__bf16)r0, (__bf16)r1, (__bf16)r2, (__bf16)r3};
          const nbf16x4 l = {(__bf16)(r0 - (float)m[0]), (__bf16)(r1 - (float)m[1]), (__bf16)(r2 - (float)m[2]), (__bf16)(r3 - (float)m[3])};
          __bf16* d = split + row * ld_split + 4 * c4;
          *reinterpret_cast<nbf16x4*>(d) = h;
          *reinterpret_cast<nbf16x4*>(d + img) = m;
          *reinterpret_cast<nbf16x4*>(d + 2 * img) = l;
        }
      }
    }
  }
}

__global__ __launch_bounds__(256) void ln_stats_scalar_kernel(const float* __restrict__ x, int rows, int C, float eps,
                                                              float* __restrict__ stats, float guard, int* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = x + row * C;
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int c = lane + 64 * i;
    v[i] = c < C ? p[c] : 0.f;
    s += v[i];
  }
  s = wave_sum(s);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int c = lane + 64 * i;
    float d = c < C ? v[i] - mean : 0.f;
    q = fmaf(d, d, q);
  }
  q = wave_sum(q);
  if (lane == 0) {
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
    stats[2 * row] = mean;
    stats[2 * row + 1] = rstd;
    if (flag && !(fabsf(mean) * rstd <= guard)) *flag = 1;
  }
}

// y = act(x*scale + shift) over the channel concat x0|x1 -> one contiguous NHWC tensor (float4 lanes)
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ x0, int c0, const float* __restrict__ x1,
                                                       int c1, const float* __restrict__ coef, float* __restrict__ y,
                                                       int hw, int silu, long long total4) {
  const int C = c0 + c1;
  const int c4n = C / 4;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / c4n;
    const int c = (int)(i - row * c4n) * 4;
    const int n = (int)(row / hw);
    const float4 v = c < c0 ? *reinterpret_cast<const float4*>(x0 + row * c0 + c)
                            : *reinterpret_cast<const float4*>(x1 + row * c1 + (c - c0));
    const float* cf = coef + ((long long)n * 2) * C + c;
    const float4 sc = *reinterpret_cast<const float4*>(cf);
    const float4 sh = *reinterpret_cast<const float4*>(cf + C);
    float4 o;
    o.x = fmaf(v.x, sc.x, sh.x); o.y = fmaf(v.y, sc.y, sh.y); o.z = fmaf(v.z, sc.z, sh.z); o.w = fmaf(v.w, sc.w, sh.w);
    if (silu) { o.x = silu_f(o.x); o.y = silu_f(o.y); o.z = silu_f(o.z); o.w = silu_f(o.w); }
    *reinterpret_cast<float4*>(y + row * C + c) = o;
  }
}

// resblock_updown (openaimodel.py:207-216,256-261): the parameter-free Upsample / Downsample of a ResBlock(up=True / down=True), NHWC.
// up: y[n][2h][2w][c] = x[n][h][w][c] (F.interpolate(scale_factor=2, mode="nearest"), :110-118); down: y[n][h][w][c] = mean of the
// 2x2 block of x[n][2h][2w][c] (avg_pool2d(2, 2), :143-160; summed row-major like ATen's loop, then / 4).  (h, w) is the SMALLER grid
// either way; one thread per 4 channels of a small-grid pixel.  HBM-bound: 5/4 of the large tensor's bytes.
__global__ __launch_bounds__(256) void resample2_kernel(const float* __restrict__ x, float* __restrict__ y, int h, int w, int C, int up,
                                                        long long total4) {
  const int c4n = C >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % c4n) * 4;
    const long long pix = i / c4n;
    const int xx = (int)(pix % w);
    const long long t = pix / w;
    const int yy = (int)(t % h);
    const long long n = t / h;
    const long long big = (((n * 2 * h + 2 * yy) * 2 * w) + 2 * xx) * C + c;      // top-left element of the 2x2 block
    const long long row = (long long)2 * w * C;
    if (up) {
      const float4 v = *reinterpret_cast<const float4*>(x + pix * C + c);
      *reinterpret_cast<float4*>(y + big) = v;
      *reinterpret_cast<float4*>(y + big + C) = v;
      *reinterpret_cast<float4*>(y + big + row) = v;
      *reinterpret_cast<float4*>(y + big + row + C) = v;
    } else {
      const float4 a = *reinterpret_cast<const float4*>(x + big), b = *reinterpret_cast<const float4*>(x + big + C);
      const float4 d = *reinterpret_cast<const float4*>(x + big + row), e = *reinterpret_cast<const float4*>(x + big + row + C);
      const float4 o = make_float4((((a.x + b.x) + d.x) + e.x) * 0.25f, (((a.y + b.y) + d.y) + e.y) * 0.25f,
                                   (((a.z + b.z) + d.z) + e.z) * 0.25f, (((a.w + b.w) + d.w) + e.w) * 0.25f);
      *reinterpret_cast<float4*>(y + pix * C + c) = o;
    }
  }
}

// use_scale_shift_norm (openaimodel.py:267-271): h = GroupNorm(h) (1 + scale) + shift with (scale, shift) = the two halves of the
// ResBlock's emb_layers output per sample -- folded into the GroupNorm coefficient planes: sc' = sc (1 + s), sh' = sh (1 + s) + t
__global__ __launch_bounds__(256) void gn_coef_film_kernel(float* __restrict__ coef, const float* __restrict__ emb, int ld, int C, int total) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int n = i / C, c = i - n * C;
  const float s = 1.0f + emb[(long long)n * ld + c], t = emb[(long long)n * ld + C + c];
  float* p = coef + ((long long)n * 2) * C + c;
  const float sc = p[0], sh = p[C];
  p[0] = sc * s;
  p[C] = fmaf(sh, s, t);
}

}  // namespace ldmk

extern "C" int ldmk_gn_coef_film(float* coef, const float* emb, int ld, int n, int c, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(coef && emb && n > 0 && c > 0 && ld >= 2 * c, "ldmk_gn_coef_film: bad args (ld=%d c=%d)", ld, c);
  hipLaunchKernelGGL(gn_coef_film_kernel, dim3((n * c + 255) / 256), dim3(256), 0, (hipStream_t)stream, coef, emb, ld, c, n * c);
  return check_launch("ldmk_gn_coef_film");
}

extern "C" int ldmk_resample2(const float* x, float* y, int n, int h, int w, int c, int up, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && y && x != y && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "ldmk_resample2: bad args (C%%4==0, out of place)");
  const long long total4 = (long long)n * h * w * (c / 4);
  long long g = (total4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(resample2_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, y, h, w, c, up ? 1 : 0, total4);
  return check_launch("ldmk_resample2");
}

extern "C" int ldmk_gn_apply(const float* x0, int c0, const float* x1, int c1, const float* coef, float* y, int n, int hw,
                             int silu, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x0 && coef && y && n > 0 && hw > 0 && c0 > 0, "ldmk_gn_apply: bad args");
  LDMK_REQUIRE((c1 == 0) == (x1 == nullptr), "ldmk_gn_apply: x1/c1 mismatch");
  LDMK_REQUIRE(c0 % 4 == 0 && c1 % 4 == 0, "ldmk_gn_apply: channel counts must be multiples of 4");
  const long long total4 = (long long)n * hw * ((c0 + c1) / 4);
  long long g = (total4 + 255) / 256;
  if (g > 8192) g = 8192;
  hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x0, c0, x1, c1, coef, y, hw,
                     silu, total4);
  return check_launch("ldmk_gn_apply");
}

extern "C" int ldmk_gn_chunks(int hw) { return (hw + ldmk::GN_PIX - 1) / ldmk::GN_PIX; }

extern "C" int ldmk_gn_partial(const float* x, int c, int n, int hw, float* partial, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && partial && c > 0 && n > 0 && hw > 0, "ldmk_gn_partial: bad args");
  const int chunks = ldmk_gn_chunks(hw);
  hipLaunchKernelGGL(gn_partial_kernel, dim3(chunks, n), dim3(256), 0, (hipStream_t)stream, x, c, hw, chunks, partial);
  return check_launch("ldmk_gn_partial");
}

extern "C" int ldmk_gn_finalize(const float* partial0, int c0, const float* partial1, int c1, int n, int hw, int groups,
                                float eps, const float* gamma, const float* beta, float* coef, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  const int C = c0 + c1;
  LDMK_REQUIRE(partial0 && c0 > 0 && n > 0 && hw > 0 && groups > 0, "ldmk_gn_finalize: bad args");
  LDMK_REQUIRE((c1 == 0) == (partial1 == nullptr), "ldmk_gn_finalize: partial1/c1 mismatch");
  LDMK_REQUIRE(C % groups == 0, "ldmk_gn_finalize: C=%d not divisible by groups=%d", C, groups);
  LDMK_REQUIRE(gamma && beta && coef, "ldmk_gn_finalize: null buffer");
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(groups, n), dim3(256), 0, (hipStream_t)stream, partial0, c0,
                     partial1, c1, hw, ldmk_gn_chunks(hw), groups, eps, gamma, beta, coef);
  return check_launch("ldmk_gn_finalize");
}

extern "C" int ldmk_gn_coef(const float* x0, int c0, const float* x1, int c1, int n, int hw, int groups, float eps,
                            const float* gamma, const float* beta, float* partial, float* coef, void* stream) {
  using namespace ldmk;
  LDMK_REQUIRE(x0 && c0 > 0 && n > 0 && hw > 0 && groups > 0, "ldmk_gn_coef: bad args");
  LDMK_REQUIRE((c1 == 0) == (x1 == nullptr), "ldmk_gn_coef: x1/c1 mismatch");
  LDMK_REQUIRE(partial, "ldmk_gn_coef: null scratch");
  float* p1 = x1 ? partial + (long long)n * ldmk_gn_chunks(hw) * c0 * 3 : nullptr;
  int rc = ldmk_gn_partial(x0, c0, n, hw, partial, stream);
  if (rc == 0 && x1) rc = ldmk_gn_partial(x1, c1, n, hw, p1, stream);
  if (rc == 0) rc = ldmk_gn_finalize(partial, c0, p1, c1, n, hw, groups, eps, gamma, beta, coef, stream);
  return rc;
}

extern "C" int ldmk_ln_stats_guard(const float* x, int rows, int c, float eps, float* stats, float guard, int* flag, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && stats && rows > 0 && c > 0 && c <= 1024, "ldmk_ln_stats: bad args (C<=1024)");
  LDMK_REQUIRE(!flag || guard > 0.f, "ldmk_ln_stats_guard: guard=%g must be positive", (double)guard);
  if (c % 4 == 0)
    hipLaunchKernelGGL(ln_stats_kernel<false>, dim3((rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, x, rows, c, eps, stats,
                       (__bf16*)nullptr, 0, guard, flag);
  else
    hipLaunchKernelGGL(ln_stats_scalar_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, rows, c, eps, stats,
                       guard, flag);
  return check_launch("ldmk_ln_stats");
}

extern "C" int ldmk_ln_stats(const float* x, int rows, int c, float eps, float* stats, void* stream) {
  return ldmk_ln_stats_guard(x, rows, c, eps, stats, 0.f, nullptr, stream);
}

extern "C" int ldmk_ln_stats_split(const float* x, int rows, int c, float eps, float* stats, void* split, int ld_split, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && stats && split && rows > 0 && c > 0 && c <= 1024 && c % 4 == 0, "ldmk_ln_stats_split: bad args (C<=1024, C%%4==0)");
  LDMK_REQUIRE(ld_split >= c && ld_split % 8 == 0, "ldmk_ln_stats_split: ld_split=%d must be >= C and a multiple of 8", ld_split);
  hipLaunchKernelGGL(ln_stats_kernel<true>, dim3((rows + 7) / 8), dim3(256), 0, (hipStream_t)stream, x, rows, c, eps, stats,
                     reinterpret_cast<__bf16*>(split), ld_split, 0.f, (int*)nullptr);
  return check_launch("ldmk_ln_stats_split");
}
