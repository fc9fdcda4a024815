// Winograd F(2x2, 3x3) transforms around a batched igemm: a stride-1, pad-1 3x3 convolution as
//     V_p = (B^T d B)_p  ->  M_p = V_p U_p  (16 GEMMs of K = C_in on the matrix cores, ldmk_igemm batch = 16)  ->  Y = A^T M A
// with 4 multiplications per output and input channel instead of 9 (Lavin & Gray 2016, the correlation form nn.Conv2d
// computes: openaimodel.py:204,230; model.py:95-129).  Used where the transforms' traffic (4x the activation, in and out)
// is small next to the matrix time saved: the 640-channel ResBlock convolutions.
//
//   B^T = | 1  0 -1  0 |   G = | 1    0    0  |   A^T = | 1  1  1  0 |
//         | 0  1  1  0 |       | 1/2  1/2  1/2|         | 0  1 -1 -1 |
//         | 0 -1  1  0 |       | 1/2 -1/2  1/2|
//         | 0  1  0 -1 |       | 0    0    1  |
//
// The input transform also applies the GroupNorm(+SiLU) that precedes the convolution (it replaces the gn_apply pass: the
// zero padding is a padding of the ACTIVATED tensor, so out-of-image taps stay 0), and reads the channel concat of two
// tensors; the output transform takes over the convolution's epilogue: bias, per-sample (timestep-embedding) vector,
// residual, and the GroupNorm partial records of the result for the next layer.
#include "ldmk_common.h"

namespace ldmk {

// thread <-> (tile, 4 channels); tiles are the 2x2 output blocks in (sample, ty, tx) order; V is [16][tiles][C]
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x0, int c0, const float* __restrict__ x1,
                                                         int c1, const float* __restrict__ coef, int silu, int H, int W,
                                                         long long tiles, float* __restrict__ V) {
  const int C = c0 + c1, c4n = C >> 2;
  const int tw = W >> 1, th = H >> 1;
  const long long total = tiles * c4n;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long t = idx / c4n;
    const int c = (int)(idx - t * c4n) * 4;
    const int tx = (int)(t % tw);
    const long long t2 = t / tw;
    const int ty = (int)(t2 % th), n = (int)(t2 / th);
    const float* src = c < c0 ? x0 + (long long)n * H * W * c0 + c : x1 + (long long)n * H * W * c1 + (c - c0);
    const int cs = c < c0 ? c0 : c1;
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (coef) {
      const float* cf = coef + ((long long)n * 2) * C + c;
      sc = *reinterpret_cast<const float4*>(cf);
      sh = *reinterpret_cast<const float4*>(cf + C);
    }
    float4 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = 2 * ty - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = 2 * tx - 1 + j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < H && x >= 0 && x < W) {
          v = *reinterpret_cast<const float4*>(src + ((long long)y * W + x) * cs);
          if (coef) { v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w); }
          if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        }
        d[i][j] = v;
      }
    }
    auto sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
    auto add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
    float4 r[4][4];                     // B^T d
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      r[0][j] = sub(d[0][j], d[2][j]);
      r[1][j] = add(d[1][j], d[2][j]);
      r[2][j] = sub(d[2][j], d[1][j]);
      r[3][j] = sub(d[1][j], d[3][j]);
    }
    float* dst = V + t * C + c;
    const long long ps = tiles * C;     // stride between transform positions
#pragma unroll
    for (int i = 0; i < 4; ++i) {       // (B^T d) B
      *reinterpret_cast<float4*>(dst + (4 * i + 0) * ps) = sub(r[i][0], r[i][2]);
      *reinterpret_cast<float4*>(dst + (4 * i + 1) * ps) = add(r[i][1], r[i][2]);
      *reinterpret_cast<float4*>(dst + (4 * i + 2) * ps) = sub(r[i][2], r[i][1]);
      *reinterpret_cast<float4*>(dst + (4 * i + 3) * ps) = sub(r[i][1], r[i][3]);
    }
  }
}

// The same transform writing V in the PS layout (include/ldmk.h; csrc/igemm_ps.hip): 16 position planes, each the PS image of a
// [tiles][C] matrix (three bf16 planes of the exact split, MFMA-operand order).  thread <-> (tile, 8 channels): lanes 0-31 of a
// wave are 32 consecutive tiles at channels 16 s .. 16 s + 7, lanes 32-63 the same tiles at 16 s + 8 .. 16 s + 15, so every store
// instruction writes ONE whole fragment plane, 1 KiB contiguous; a workgroup = 32 tiles x 4 k-slabs.  Same arithmetic (and
// order) as wino_input_kernel: V is bit for bit the same matrix, split exactly.
// PL = 2: the F16X2 form of the layout (two fp16 planes of 2^6 x per 2-KiB unit; an element of 1000 or more raises *range_flag).
typedef __bf16 wbf16x8_ __attribute__((ext_vector_type(8)));
typedef _Float16 wf16x8_ __attribute__((ext_vector_type(8)));
template <int PL = 3>
__device__ __forceinline__ void wino_store_ps(unsigned char* d, const float4& a, const float4& b, int* range_flag = nullptr) {
  float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  if constexpr (PL == 2) {
    wf16x8_ h, l;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      bad |= (__float_as_uint(v[e]) & 0x7fffffffu) >= 0x447a0000u;
      const float s = h2_clamp(v[e]) * 64.f;      // (saturated: ldmk_common.h)
      h[e] = (_Float16)s;
      l[e] = (_Float16)(s - (float)h[e]);
    }
    if (bad) *range_flag = 1;
    *reinterpret_cast<wf16x8_*>(d) = h;
    *reinterpret_cast<wf16x8_*>(d + 1024) = l;
    return;
  }
  wbf16x8_ h, m, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    h[e] = (__bf16)v[e];
    const float r = v[e] - (float)h[e];
    m[e] = (__bf16)r;
    l[e] = (__bf16)(r - (float)m[e]);
  }
  *reinterpret_cast<wbf16x8_*>(d) = h;
  *reinterpret_cast<wbf16x8_*>(d + 1024) = m;
  *reinterpret_cast<wbf16x8_*>(d + 2048) = l;
}

template <int PL>
__global__ __launch_bounds__(256) void wino_input_ps_kernel(const float* __restrict__ x0, int c0, const float* __restrict__ x1, int c1,
                                                            const float* __restrict__ coef, int silu, int H, int W, long long tiles,
                                                            unsigned char* __restrict__ V, long long plane_bytes, int* __restrict__ range_flag) {
  const int C = c0 + c1, Kb = C >> 4;
  const int tw = W >> 1, th = H >> 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int slab = blockIdx.y * 4 + wave;
  if (slab >= Kb) return;
  const long long t = (long long)blockIdx.x * 32 + r;
  const int c = slab * 16 + hh * 8;
  float4 d[4][4][2];
  const bool tok = t < tiles;
  if (tok) {
    const int tx = (int)(t % tw);
    const long long t2 = t / tw;
    const int ty = (int)(t2 % th), n = (int)(t2 / th);
    const float* src = c < c0 ? x0 + (long long)n * H * W * c0 + c : x1 + (long long)n * H * W * c1 + (c - c0);
    const int cs = c < c0 ? c0 : c1;
    float4 sc[2] = {make_float4(1.f, 1.f, 1.f, 1.f), make_float4(1.f, 1.f, 1.f, 1.f)};
    float4 sh[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
    if (coef) {
      const float* cf = coef + ((long long)n * 2) * C + c;
      sc[0] = *reinterpret_cast<const float4*>(cf); sc[1] = *reinterpret_cast<const float4*>(cf + 4);
      sh[0] = *reinterpret_cast<const float4*>(cf + C); sh[1] = *reinterpret_cast<const float4*>(cf + C + 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = 2 * ty - 1 + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int x = 2 * tx - 1 + j;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
          if (in) {
            v = *reinterpret_cast<const float4*>(src + ((long long)y * W + x) * cs + 4 * e);
            if (coef) { v.x = fmaf(v.x, sc[e].x, sh[e].x); v.y = fmaf(v.y, sc[e].y, sh[e].y); v.z = fmaf(v.z, sc[e].z, sh[e].z); v.w = fmaf(v.w, sc[e].w, sh[e].w); }
            if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
          }
          d[i][j][e] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j][0] = d[i][j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  auto sub = [](float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); };
  auto add = [](float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); };
  unsigned char* dst = V + ((long long)blockIdx.x * Kb + slab) * (PL * 1024) + lane * 16;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float4 rr[4][2];                    // row i of B^T d
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rr[j][e] = i == 0 ? sub(d[0][j][e], d[2][j][e]) : i == 1 ? add(d[1][j][e], d[2][j][e]) : i == 2 ? sub(d[2][j][e], d[1][j][e])
                                                                                                          : sub(d[1][j][e], d[3][j][e]);
      }
    // (B^T d) B, positions 4 i .. 4 i + 3
    wino_store_ps<PL>(dst + (4 * i + 0) * plane_bytes, sub(rr[0][0], rr[2][0]), sub(rr[0][1], rr[2][1]), range_flag);
    wino_store_ps<PL>(dst + (4 * i + 1) * plane_bytes, add(rr[1][0], rr[2][0]), add(rr[1][1], rr[2][1]), range_flag);
    wino_store_ps<PL>(dst + (4 * i + 2) * plane_bytes, sub(rr[2][0], rr[1][0]), sub(rr[2][1], rr[1][1]), range_flag);
    wino_store_ps<PL>(dst + (4 * i + 3) * plane_bytes, sub(rr[1][0], rr[3][0]), sub(rr[1][1], rr[3][1]), range_flag);
  }
}

// One workgroup = (sample, band of 2R image rows); thread <-> channel (coalesced over N), walking the band's tiles left to
// right.  The band is a whole number of 32-pixel GroupNorm chunks (W = 8: R = 2; W >= 16: R = 1), so the partial records of
// the result (same records as gn_partial_kernel: shift, sum, sum of squares of the chunk) are complete per workgroup.
constexpr int WINO_MAX_CHUNKS = 8;      // Winograd output: W <= 128; upsample scatter: low-resolution W <= 64
__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ Mb, const float* __restrict__ bias,
                                                          const float* __restrict__ bvec, int bvec_ld,
                                                          const float* __restrict__ res, float* __restrict__ out,
                                                          float* __restrict__ stats, int H, int W, int N, int R,
                                                          long long tiles) {
  const int tw = W >> 1, th = H >> 1;
  const int bands = th / R;
  const int n = blockIdx.x / bands, band = blockIdx.x - n * bands;
  const long long ps = tiles * N;
  const int nchunks = (2 * R * W) >> 5;
  const int c = blockIdx.y * blockDim.x + threadIdx.x;      // grid.y covers the channels
  if (c < N) {
    const float add0 = (bias ? bias[c] : 0.f) + (bvec ? bvec[(long long)n * bvec_ld + c] : 0.f);
    float shift[WINO_MAX_CHUNKS], sm[WINO_MAX_CHUNKS], sq[WINO_MAX_CHUNKS];
    bool have[WINO_MAX_CHUNKS];
#pragma unroll
    for (int k = 0; k < WINO_MAX_CHUNKS; ++k) { shift[k] = 0.f; sm[k] = 0.f; sq[k] = 0.f; have[k] = false; }
    for (int r = 0; r < R; ++r) {
      const int ty = band * R + r;
      for (int tx = 0; tx < tw; ++tx) {
        const long long t = ((long long)n * th + ty) * tw + tx;
        const float* mp = Mb + t * N + c;
        float m[16];
#pragma unroll
        for (int p = 0; p < 16; ++p) m[p] = mp[p * ps];
        float s0[4], s1[4];                 // A^T m
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0[j] = (m[j] + m[4 + j]) + m[8 + j];
          s1[j] = (m[4 + j] - m[8 + j]) - m[12 + j];
        }
        float y[2][2];
        y[0][0] = (s0[0] + s0[1]) + s0[2];
        y[0][1] = (s0[1] - s0[2]) - s0[3];
        y[1][0] = (s1[0] + s1[1]) + s1[2];
        y[1][1] = (s1[1] - s1[2]) - s1[3];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int yy = 2 * ty + i, xx = 2 * tx + j;
            const long long o = (((long long)n * H + yy) * W + xx) * N + c;
            float v = y[i][j] + add0;
            if (res) v += res[o];
            out[o] = v;
            if (stats) {
              const int k = (((2 * r + i) * W) + xx) >> 5;     // chunk of this pixel inside the band
#pragma unroll
              for (int q = 0; q < WINO_MAX_CHUNKS; ++q)
                if (q == k) {
                  if (!have[q]) { shift[q] = v; have[q] = true; }
                  const float dv = v - shift[q];
                  sm[q] += dv;
                  sq[q] = fmaf(dv, dv, sq[q]);
                }
            }
          }
      }
    }
    if (stats) {
      const long long chunk0 = ((long long)n * H * W + (long long)band * 2 * R * W) >> 5;
#pragma unroll
      for (int q = 0; q < WINO_MAX_CHUNKS; ++q)
        if (q < nchunks) {
          float* d = stats + ((chunk0 + q) * N + c) * 3;
          d[0] = shift[q]; d[1] = sm[q]; d[2] = sq[q];
        }
    }
  }
}

}  // namespace ldmk

extern "C" long long ldmk_winograd_tiles(int n, int h, int w) { return (long long)n * (h / 2) * (w / 2); }

extern "C" int ldmk_winograd_input(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h,
                                   int w, float* v, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x0 && v && n > 0 && c0 > 0, "ldmk_winograd_input: bad args");
  LDMK_REQUIRE((c1 == 0) == (x1 == nullptr), "ldmk_winograd_input: x1/c1 mismatch");
  LDMK_REQUIRE(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "ldmk_winograd_input: H=%d W=%d must be even", h, w);
  LDMK_REQUIRE(c0 % 4 == 0 && c1 % 4 == 0, "ldmk_winograd_input: channel counts must be multiples of 4");
  const long long tiles = ldmk_winograd_tiles(n, h, w);
  const long long total = tiles * ((c0 + c1) / 4);
  long long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(wino_input_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x0, c0, x1, c1, coef, silu, h, w,
                     tiles, v);
  return check_launch("ldmk_winograd_input");
}

extern "C" long long ldmk_ps_bytes(int rows, int k);
extern "C" long long ldmk_ps_bytes_h2(int rows, int k);

static int winograd_input_ps_any(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h, int w, void* v_ps,
                                 int* range_flag, void* stream);

extern "C" int ldmk_winograd_input_ps(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h,
                                      int w, void* v_ps, void* stream) {
  return winograd_input_ps_any(x0, c0, x1, c1, coef, silu, n, h, w, v_ps, nullptr, stream);
}

extern "C" int ldmk_winograd_input_ps_h2(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h,
                                         int w, void* v_ps, int* range_flag, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(range_flag != nullptr, "ldmk_winograd_input_ps_h2: range_flag");
  return winograd_input_ps_any(x0, c0, x1, c1, coef, silu, n, h, w, v_ps, range_flag, stream);
}

static int winograd_input_ps_any(const float* x0, int c0, const float* x1, int c1, const float* coef, int silu, int n, int h, int w, void* v_ps,
                                 int* range_flag, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x0 && v_ps && n > 0 && c0 > 0, "ldmk_winograd_input_ps: bad args");
  LDMK_REQUIRE((c1 == 0) == (x1 == nullptr), "ldmk_winograd_input_ps: x1/c1 mismatch");
  LDMK_REQUIRE(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "ldmk_winograd_input_ps: H=%d W=%d must be even", h, w);
  LDMK_REQUIRE(c0 % 16 == 0 && c1 % 16 == 0, "ldmk_winograd_input_ps: channel counts must be multiples of 16 (k-slabs of the PS layout)");
  const long long tiles = ldmk_winograd_tiles(n, h, w);
  LDMK_REQUIRE(tiles < (1LL << 31), "ldmk_winograd_input_ps: too many tiles");
  const int C = c0 + c1;
  const dim3 grid((unsigned)((tiles + 31) / 32), (C / 16 + 3) / 4);
  if (range_flag)      // the F16X2 form: two fp16 planes of 2^6 V
    hipLaunchKernelGGL(wino_input_ps_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x0, c0, x1, c1, coef, silu, h, w, tiles,
                       reinterpret_cast<unsigned char*>(v_ps), ldmk_ps_bytes_h2((int)tiles, C), range_flag);
  else
    hipLaunchKernelGGL(wino_input_ps_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x0, c0, x1, c1, coef, silu, h, w, tiles,
                       reinterpret_cast<unsigned char*>(v_ps), ldmk_ps_bytes((int)tiles, C), range_flag);
  return check_launch("ldmk_winograd_input_ps");
}

extern "C" int ldmk_winograd_output(const float* m, const float* bias, const float* batch_vec, int batch_vec_ld,
                                    const float* residual, float* out, float* stats_out, int n, int h, int w, int cout,
                                    void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(m && out && n > 0 && cout > 0, "ldmk_winograd_output: bad args");
  LDMK_REQUIRE(h >= 2 && w >= 2 && h % 2 == 0 && w % 2 == 0, "ldmk_winograd_output: H=%d W=%d must be even", h, w);
  int R = 1;
  if (stats_out) {
    // a band of 2R image rows must be a whole number of 32-pixel chunks and at most WINO_MAX_CHUNKS of them
    while ((2 * R * w) % 32 != 0 && R < h / 2) R *= 2;
    LDMK_REQUIRE((2 * R * w) % 32 == 0 && (h / 2) % R == 0 && (2 * R * w) / 32 <= WINO_MAX_CHUNKS && (h * w) % 32 == 0,
                 "ldmk_winograd_output: stats_out needs bands of whole 32-pixel chunks (H=%d W=%d)", h, w);
  }
  const long long tiles = ldmk_winograd_tiles(n, h, w);
  hipLaunchKernelGGL(wino_output_kernel, dim3(n * ((h / 2) / R), (cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, m, bias, batch_vec,
                     batch_vec_ld, residual, out, stats_out, h, w, cout, R, tiles);
  return check_launch("ldmk_winograd_output");
}

// ---------------------------------------------------------------------------------------------
// Nearest-x2 upsampling followed by a 3x3 convolution (openaimodel.py:107-118, model.py:45-57) as FOUR 2x2-tap convolutions
// on the low-resolution input, one per output parity (a, b): the upsampled rows 2y+a-1 .. 2y+a+1 are the low-resolution rows
// {y-1, y, y} (a = 0) or {y, y, y+1} (a = 1), so the three row taps collapse to two with weights (w0, w1+w2) or (w0+w1, w2),
// and the same in x -- 4 instead of 9 multiplications per output and input channel, no approximation.
//   ldmk_upconv_gather : A[4 phases][n h w][4 taps x C], tap (i, j) of phase (a, b) = x[y + a - 1 + i][x + b - 1 + j] (zeros outside)
//   ldmk_igemm         : batch = 4, M = n h w, K = 4 C, N = cout, weights from ops.pack_upconv
//   ldmk_upconv_scatter: out[n][2h][2w][cout] = phase planes interleaved + bias, + GroupNorm partial records of the result
namespace ldmk {

__global__ __launch_bounds__(256) void upconv_gather_kernel(const float* __restrict__ x, int C, int H, int W, long long pix,
                                                            float* __restrict__ A) {
  const int c4n = C >> 2;
  const long long total = pix * c4n;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
    const long long p = idx / c4n;
    const int c = (int)(idx - p * c4n) * 4;
    const int xx = (int)(p % W);
    const long long p2 = p / W;
    const int yy = (int)(p2 % H), n = (int)(p2 / H);
    float4 d[3][3];                     // the 3x3 low-resolution neighbourhood, zeros outside the image
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int y = yy - 1 + i, x_ = xx - 1 + j;
        d[i][j] = (y >= 0 && y < H && x_ >= 0 && x_ < W)
                      ? *reinterpret_cast<const float4*>(x + (((long long)n * H + y) * W + x_) * C + c)
                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    const long long plane = pix * 4 * C;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        float* dst = A + (long long)(2 * a + b) * plane + p * 4 * C + c;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) *reinterpret_cast<float4*>(dst + (2 * i + j) * C) = d[a + i][b + j];
      }
  }
}

// the gather writing its four phase operands in the PS layout (csrc/igemm_ps.hip): per phase the PS image of [pix][4 C]; thread <->
// (pixel, 8 channels) with the lane order of wino_input_ps_kernel, so every store is one whole 1-KiB fragment plane
template <int PL>
__global__ __launch_bounds__(256) void upconv_gather_ps_kernel(const float* __restrict__ x, int C, int H, int W, long long pix,
                                                               unsigned char* __restrict__ A, long long plane_bytes, int* __restrict__ range_flag) {
  const int Kc = C >> 4;                 // k-slabs per tap
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, hh = lane >> 5;
  const int slab = blockIdx.y * 4 + wave;
  if (slab >= Kc) return;
  const long long p = (long long)blockIdx.x * 32 + r;
  const int c = slab * 16 + hh * 8;
  float4 d[3][3][2];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) d[i][j][0] = d[i][j][1] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p < pix) {
    const int xx = (int)(p % W);
    const long long p2 = p / W;
    const int yy = (int)(p2 % H), n = (int)(p2 / H);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int y = yy - 1 + i, x_ = xx - 1 + j;
        if (y >= 0 && y < H && x_ >= 0 && x_ < W) {
          const float* s = x + (((long long)n * H + y) * W + x_) * C + c;
          d[i][j][0] = *reinterpret_cast<const float4*>(s);
          d[i][j][1] = *reinterpret_cast<const float4*>(s + 4);
        }
      }
  }
  const int Kb = 4 * Kc;                  // k-slabs per row of the [pix][4 C] operand
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      unsigned char* dst = A + (long long)(2 * a + b) * plane_bytes + ((long long)blockIdx.x * Kb + slab) * (PL * 1024) + lane * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          wino_store_ps<PL>(dst + (long long)(2 * i + j) * Kc * (PL * 1024), d[a + i][b + j][0], d[a + i][b + j][1], range_flag);
    }
}

// workgroup = (sample, low-resolution row): two output rows of 2W pixels = 2W/16 whole 32-pixel chunks; thread <-> channel
__global__ __launch_bounds__(256) void upconv_scatter_kernel(const float* __restrict__ Pm, const float* __restrict__ bias,
                                                             float* __restrict__ out, float* __restrict__ stats, int H, int W,
                                                             int N, long long pix) {
  const int n = blockIdx.x / H, yy = blockIdx.x - n * H;
  const int c = blockIdx.y * blockDim.x + threadIdx.x;
  if (c >= N) return;
  const int W2 = 2 * W;
  const float b0 = bias ? bias[c] : 0.f;
  const long long plane = pix * N;
  float shift[WINO_MAX_CHUNKS], sm[WINO_MAX_CHUNKS], sq[WINO_MAX_CHUNKS];
  bool have[WINO_MAX_CHUNKS];
#pragma unroll
  for (int k = 0; k < WINO_MAX_CHUNKS; ++k) { shift[k] = 0.f; sm[k] = 0.f; sq[k] = 0.f; have[k] = false; }
  for (int xx = 0; xx < W; ++xx) {
    const long long p = ((long long)n * H + yy) * W + xx;
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = Pm[q * plane + p * N + c] + b0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ox = 2 * xx + b;
        const float val = v[2 * a + b];
        out[(((long long)n * 2 * H + 2 * yy + a) * W2 + ox) * N + c] = val;
        if (stats) {
          const int k = (a * W2 + ox) >> 5;
#pragma unroll
          for (int q = 0; q < WINO_MAX_CHUNKS; ++q)
            if (q == k) {
              if (!have[q]) { shift[q] = val; have[q] = true; }
              const float dv = val - shift[q];
              sm[q] += dv;
              sq[q] = fmaf(dv, dv, sq[q]);
            }
        }
      }
  }
  if (stats) {
    const int nchunks = (2 * W2) >> 5;
    const long long chunk0 = ((long long)n * 4 * H * W + (long long)yy * 2 * W2) >> 5;
#pragma unroll
    for (int q = 0; q < WINO_MAX_CHUNKS; ++q)
      if (q < nchunks) {
        float* d = stats + ((chunk0 + q) * N + c) * 3;
        d[0] = shift[q]; d[1] = sm[q]; d[2] = sq[q];
      }
  }
}

}  // namespace ldmk

extern "C" int ldmk_upconv_gather(const float* x, int c, int n, int h, int w, float* a, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && a && n > 0 && h > 0 && w > 0 && c > 0 && c % 4 == 0, "ldmk_upconv_gather: bad args");
  const long long pix = (long long)n * h * w, total = pix * (c / 4);
  long long g = (total + 255) / 256;
  if (g > 16384) g = 16384;
  hipLaunchKernelGGL(upconv_gather_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, x, c, h, w, pix, a);
  return check_launch("ldmk_upconv_gather");
}

static int upconv_gather_ps_any(const float* x, int c, int n, int h, int w, void* a_ps, int* range_flag, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && a_ps && n > 0 && h > 0 && w > 0 && c > 0 && c % 16 == 0, "ldmk_upconv_gather_ps: bad args (C a multiple of 16)");
  const long long pix = (long long)n * h * w;
  LDMK_REQUIRE(pix < (1LL << 31), "ldmk_upconv_gather_ps: too many pixels");
  const dim3 grid((unsigned)((pix + 31) / 32), (c / 16 + 3) / 4);
  if (range_flag)
    hipLaunchKernelGGL(upconv_gather_ps_kernel<2>, grid, dim3(256), 0, (hipStream_t)stream, x, c, h, w, pix, reinterpret_cast<unsigned char*>(a_ps),
                       ldmk_ps_bytes_h2((int)pix, 4 * c), range_flag);
  else
    hipLaunchKernelGGL(upconv_gather_ps_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, x, c, h, w, pix, reinterpret_cast<unsigned char*>(a_ps),
                       ldmk_ps_bytes((int)pix, 4 * c), range_flag);
  return check_launch("ldmk_upconv_gather_ps");
}

extern "C" int ldmk_upconv_gather_ps(const float* x, int c, int n, int h, int w, void* a_ps, void* stream) {
  return upconv_gather_ps_any(x, c, n, h, w, a_ps, nullptr, stream);
}

extern "C" int ldmk_upconv_gather_ps_h2(const float* x, int c, int n, int h, int w, void* a_ps, int* range_flag, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(range_flag != nullptr, "ldmk_upconv_gather_ps_h2: range_flag");
  return upconv_gather_ps_any(x, c, n, h, w, a_ps, range_flag, stream);
}

extern "C" int ldmk_upconv_scatter(const float* planes, const float* bias, float* out, float* stats_out, int n, int h, int w,
                                   int cout, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(planes && out && n > 0 && h > 0 && w > 0 && cout > 0, "ldmk_upconv_scatter: bad args");
  if (stats_out)
    LDMK_REQUIRE((2 * w) % 16 == 0 && (4 * w) / 32 <= WINO_MAX_CHUNKS && (4 * w) % 32 == 0,
                 "ldmk_upconv_scatter: stats_out needs two output rows (2 x %d pixels) to be 1..%d whole 32-pixel chunks", 2 * w,
                 WINO_MAX_CHUNKS);
  hipLaunchKernelGGL(upconv_scatter_kernel, dim3(n * h, (cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, planes, bias, out,
                     stats_out, h, w, cout, (long long)n * h * w);
  return check_launch("ldmk_upconv_scatter");
}
