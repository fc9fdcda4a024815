// Small / bandwidth-bound kernels of the sampling path: per-sample dense layers, timestep
// embedding, narrow-channel boundary convolutions, sampler updates, VQ lookup, layout helpers.
#include "ldmk_common.h"
#include <string.h>

namespace ldmk {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---------------------------------------------------------------------------------------------
// out[b][n] = sum_k act(x[b][k]) * W[k][n] + bias[n]   (weight-bandwidth bound: W is read once).
// Workgroup = 64 output columns x DS_NW K-parts (one wave each, coalesced 256-B rows of W); up to
// DS_ROWS batch rows accumulate in registers against x rows broadcast from LDS; the K-part
// partials are summed through LDS in a fixed order.  A wave keeps 16 weight rows in flight: the kernel is a chain of
// memory round trips (round 2: 4 parts x 8 loads = 20 round trips for the 640 x 7040 emb_layers matrix, 21 us per call
// at batch 1; now 8 parts x 16 loads = 5).
constexpr int DS_ROWS = 16;
constexpr int DS_KT = 512;
constexpr int DS_NW = 8;
constexpr int DS_LD = 16;
__global__ __launch_bounds__(64 * DS_NW) void dense_small_kernel(const float* __restrict__ x, int ldx,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int ldo, int rows, int K, int N,
                                                          int silu_in) {
  __shared__ __attribute__((aligned(16))) float xs[DS_KT][DS_ROWS];   // [k][row]: one k = 4 broadcast b128 reads
  __shared__ float red[DS_NW - 1][DS_ROWS][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + lane;
  const int r0 = blockIdx.y * DS_ROWS;
  const int nr = min(DS_ROWS, rows - r0);
  float acc[DS_ROWS];
#pragma unroll
  for (int r = 0; r < DS_ROWS; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += DS_KT) {
    const int kn = min(DS_KT, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < DS_ROWS * DS_KT; i += 64 * DS_NW) {
      const int r = i / DS_KT, kk = i - r * DS_KT;          // consecutive threads read consecutive k of one row
      float v = 0.f;
      if (r < nr && kk < kn) {
        v = x[(long long)(r0 + r) * ldx + k0 + kk];
        if (silu_in) v = silu_f(v);
      }
      xs[kk][r] = v;
    }
    __syncthreads();
    if (n < N) {
      const int q = (kn + DS_NW - 1) / DS_NW;
      const int kb = wave * q, ke = min(kn, kb + q);
      const float* wp = w + (long long)(k0 + kb) * N + n;
      for (int kk = kb; kk < ke; kk += DS_LD) {
        float wv[DS_LD];                   // DS_LD independent weight loads in flight per lane before any use
#pragma unroll
        for (int u = 0; u < DS_LD; ++u) wv[u] = (kk + u < ke) ? wp[(long long)u * N] : 0.f;
        wp += (long long)DS_LD * N;
#pragma unroll
        for (int u = 0; u < DS_LD; ++u) {
          if (kk + u >= ke) break;
          const float4* xr = reinterpret_cast<const float4*>(&xs[kk + u][0]);
#pragma unroll
          for (int r4 = 0; r4 < DS_ROWS / 4; ++r4) {
            const float4 xv = xr[r4];
            acc[4 * r4] = fmaf(xv.x, wv[u], acc[4 * r4]);
            acc[4 * r4 + 1] = fmaf(xv.y, wv[u], acc[4 * r4 + 1]);
            acc[4 * r4 + 2] = fmaf(xv.z, wv[u], acc[4 * r4 + 2]);
            acc[4 * r4 + 3] = fmaf(xv.w, wv[u], acc[4 * r4 + 3]);
          }
        }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < DS_ROWS; ++r) red[wave - 1][r][lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0 && n < N) {
    const float bv = bias ? bias[n] : 0.f;
    for (int r = 0; r < nr; ++r) {
      float v = acc[r];
#pragma unroll
      for (int q = 0; q < DS_NW - 1; ++q) v += red[q][r][lane];
      out[(long long)(r0 + r) * ldo + n] = v + bv;
    }
  }
}

// The same product with 16-byte weight loads (N % 4 == 0): a lane owns 4 columns, the four 16-lane groups of a wave take four
// consecutive K rows, so one load instruction moves 1 KB per wave (4x the scalar form) and a wave keeps 8 of them in flight.
// At batch 16 the scalar form had 160 workgroups x 32 KB in flight on the 52 MB emb_layers matrix (latency-bound, ~0.9 TB/s).
// ROWS = batch rows accumulated per workgroup (4 for the batch-1/2 route: a quarter of the FMAs).
constexpr int DS4_LD = 8;
template <int ROWS>
__global__ __launch_bounds__(64 * DS_NW) void dense_small4_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ w,
                                                                   const float* __restrict__ bias, float* __restrict__ out, int ldo,
                                                                   int rows, int K, int N, int silu_in) {
  __shared__ __attribute__((aligned(16))) float xs[DS_KT][ROWS];
  __shared__ __attribute__((aligned(16))) float red[DS_NW - 1][ROWS][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int kq = lane >> 4, c4 = lane & 15;
  const int n = blockIdx.x * 64 + c4 * 4;
  const int r0 = blockIdx.y * ROWS;
  const int nr = min(ROWS, rows - r0);
  float acc[ROWS][4];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) acc[r][0] = acc[r][1] = acc[r][2] = acc[r][3] = 0.f;
  for (int k0 = 0; k0 < K; k0 += DS_KT) {
    const int kn = min(DS_KT, K - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < ROWS * DS_KT; i += 64 * DS_NW) {
      const int r = i / DS_KT, kk = i - r * DS_KT;
      float v = 0.f;
      if (r < nr && kk < kn) {
        v = x[(long long)(r0 + r) * ldx + k0 + kk];
        if (silu_in) v = silu_f(v);
      }
      xs[kk][r] = v;
    }
    __syncthreads();
    if (n < N) {
      const int q = ((kn + DS_NW - 1) / DS_NW + 3) & ~3;          // K rows per wave, a multiple of the 4 row groups
      const int kb = wave * q, ke = min(kn, kb + q);
      const float* wp = w + (long long)(k0 + kb + kq) * N + n;
      for (int kk = kb; kk < ke; kk += 4 * DS4_LD) {
        float4 wv[DS4_LD];
#pragma unroll
        for (int u = 0; u < DS4_LD; ++u)
          wv[u] = (kk + 4 * u + kq < ke) ? *reinterpret_cast<const float4*>(wp + (long long)(4 * u) * N) : make_float4(0.f, 0.f, 0.f, 0.f);
        wp += (long long)(4 * DS4_LD) * N;
#pragma unroll
        for (int u = 0; u < DS4_LD; ++u) {
          const float* xr = &xs[min(kk + 4 * u + kq, DS_KT - 1)][0];       // (rows past ke meet zero weights)
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const float xv = xr[r];
            acc[r][0] = fmaf(xv, wv[u].x, acc[r][0]); acc[r][1] = fmaf(xv, wv[u].y, acc[r][1]);
            acc[r][2] = fmaf(xv, wv[u].z, acc[r][2]); acc[r][3] = fmaf(xv, wv[u].w, acc[r][3]);
          }
        }
      }
    }
  }
  // the four K-row groups of the wave, then the waves: fixed order
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float v = acc[r][c];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      acc[r][c] = v;
    }
  if (wave > 0 && kq == 0) {
#pragma unroll
    for (int r = 0; r < ROWS; ++r) *reinterpret_cast<float4*>(&red[wave - 1][r][c4 * 4]) = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
  }
  __syncthreads();
  if (wave == 0 && kq == 0 && n < N) {
    const float4 bv = bias ? *reinterpret_cast<const float4*>(bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < nr; ++r) {
      float4 v = make_float4(acc[r][0], acc[r][1], acc[r][2], acc[r][3]);
#pragma unroll
      for (int qq = 0; qq < DS_NW - 1; ++qq) {
        const float4 t = *reinterpret_cast<const float4*>(&red[qq][r][c4 * 4]);
        v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
      }
      v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
      *reinterpret_cast<float4*>(out + (long long)(r0 + r) * ldo + n) = v;
    }
  }
}

__global__ void timestep_embedding_kernel(const long long* __restrict__ t, const float* __restrict__ freqs,
                                          float* __restrict__ emb, int n, int dim) {
  const int half = dim / 2;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * half) return;
  int b = idx / half, i = idx - b * half;
  // args = float(t) * freqs  (util.py:161-165); the frequency table is a host-computed constant
  float arg = (float)t[b] * freqs[i];
  // evaluated in double and rounded once: |arg| reaches ~1e3 rad where fp32 range reduction costs accuracy
  emb[(long long)b * dim + i] = (float)cos((double)arg);
  emb[(long long)b * dim + half + i] = (float)sin((double)arg);
  if ((dim & 1) && i == 0) emb[(long long)b * dim + dim - 1] = 0.f;
}

// ---------------------------------------------------------------------------------------------
// 3x3 pad-1 convolution from a narrow NCHW input (<= 16 channels, optionally the concat of two
// tensors) to a wide NHWC output (openaimodel.py:515-517, model.py:394-398,503-507).
// Workgroup = one segment of up to 64 pixels of one image row.  The 3 input rows of the segment (+halo) are staged in
// LDS once; thread <-> output channel, with that channel's 9*cin weights held in registers for the whole segment
// (the first version re-read them from L2 for every pixel: 23 KB per pixel, 1.5 GB per launch at 64x64, 123 us).
// Per pixel a thread reads the 9*cin patch values as LDS broadcasts and writes one float: a wave's store is a
// contiguous 256-B run of the NHWC row.
constexpr int CIN_MAX = 16, CIN_SEG = 64;
template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_in_kernel(const float* __restrict__ x0, int c0,
                                                         const float* __restrict__ x1, int c1,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         float* __restrict__ out, int n, int h, int wd, int cout) {
  __shared__ float patch[3][CIN][CIN_SEG + 2];
  const int segs = (wd + CIN_SEG - 1) / CIN_SEG;
  const int seg = blockIdx.x % segs, y = (blockIdx.x / segs) % h, b = blockIdx.x / (segs * h);
  const int xs = seg * CIN_SEG, np = min(CIN_SEG, wd - xs);
  const int hw = h * wd;
  for (int i = threadIdx.x; i < 3 * CIN * (CIN_SEG + 2); i += 256) {
    const int px = i % (CIN_SEG + 2), c = (i / (CIN_SEG + 2)) % CIN, dy = i / ((CIN_SEG + 2) * CIN);
    const int yy = y + dy - 1, xx = xs + px - 1;
    float v = 0.f;
    if (yy >= 0 && yy < h && xx >= 0 && xx < wd)
      v = c < c0 ? x0[((long long)b * c0 + c) * hw + yy * wd + xx] : x1[((long long)b * c1 + (c - c0)) * hw + yy * wd + xx];
    patch[dy][c][px] = v;
  }
  __syncthreads();
  float* orow = out + ((long long)b * hw + (long long)y * wd + xs) * cout;
  for (int co = threadIdx.x; co < cout; co += 256) {
    float wr[9][CIN];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int c = 0; c < CIN; ++c) wr[t][c] = w[((long long)t * CIN + c) * cout + co];
    const float bv = bias ? bias[co] : 0.f;
    for (int p0 = 0; p0 < np; p0 += 4) {         // 4 pixels per pass: every LDS value feeds up to 3 of them
      float acc[4] = {bv, bv, bv, bv};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
          float v[6];
#pragma unroll
          for (int q = 0; q < 6; ++q) v[q] = patch[dy][c][min(p0 + q, CIN_SEG + 1)];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) acc[q] = fmaf(v[q + dx], wr[dy * 3 + dx][c], acc[q]);
        }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (p0 + q < np) orow[(long long)(p0 + q) * cout + co] = acc[q];
    }
  }
}

// GroupNorm+SiLU (coef planes) -> 3x3 pad-1 conv to <= 4 channels, NHWC in -> NCHW out (openaimodel.py:682-686,
// model.py:457-459,566-568).  Workgroup = 16x16 output pixels; the 18x18 input tile is staged through LDS in 32-channel
// chunks with the normalisation and SiLU applied ONCE per element (the first version, one wave per output pixel,
// re-normalised every element for each of its 9 uses: 353 us at 16x64x64x160); thread <-> pixel, its <= 4 accumulators
// walk 9 taps x 32 channels per chunk with ds_read_b128 (pixel stride 36 floats: conflict-free for the 16 pixels a
// b128 read serves together) and wave-uniform weights.
constexpr int CO_T = 16, CO_CK = 32, CO_STR = 36;
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_out_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                          const float* __restrict__ w, const float* __restrict__ bias,
                                                          float* __restrict__ out, int n, int h, int wd, int cin) {
  __shared__ __attribute__((aligned(16))) float tile[(CO_T + 2) * (CO_T + 2) * CO_STR];
  __shared__ __attribute__((aligned(16))) float wch[9 * CO_CK * 4];      // this chunk's weights, [tap][c][4] (cout padded)
  const int tx_n = (wd + CO_T - 1) / CO_T, ty_n = (h + CO_T - 1) / CO_T;
  const int b = blockIdx.x / (tx_n * ty_n), t = blockIdx.x % (tx_n * ty_n);
  const int y0 = (t / tx_n) * CO_T, x0 = (t % tx_n) * CO_T;
  const int ty = threadIdx.x / CO_T, tx = threadIdx.x % CO_T;
  const float* sc = coef ? coef + ((long long)b * 2) * cin : nullptr;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int c0 = 0; c0 < cin; c0 += CO_CK) {
    __syncthreads();                                   // the previous chunk's reads are done
    for (int i = threadIdx.x; i < (CO_T + 2) * (CO_T + 2) * (CO_CK / 4); i += 256) {
      const int c4 = i % (CO_CK / 4), pix = i / (CO_CK / 4);
      const int yy = y0 + pix / (CO_T + 2) - 1, xx = x0 + pix % (CO_T + 2) - 1;
      const int c = c0 + 4 * c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);      // zero padding applies to the ACTIVATED tensor
      if (yy >= 0 && yy < h && xx >= 0 && xx < wd && c < cin) {
        v = *reinterpret_cast<const float4*>(x + (((long long)b * h + yy) * wd + xx) * cin + c);
        if (sc) {
          const float4 s4 = *reinterpret_cast<const float4*>(sc + c), t4 = *reinterpret_cast<const float4*>(sc + cin + c);
          v.x = silu_f(fmaf(v.x, s4.x, t4.x)); v.y = silu_f(fmaf(v.y, s4.y, t4.y));
          v.z = silu_f(fmaf(v.z, s4.z, t4.z)); v.w = silu_f(fmaf(v.w, s4.w, t4.w));
        }
      }
      *reinterpret_cast<float4*>(tile + pix * CO_STR + 4 * c4) = v;
    }
    for (int i = threadIdx.x; i < 9 * CO_CK * 4; i += 256) {
      const int co = i & 3, c = (i >> 2) % CO_CK, tap = i / (4 * CO_CK);
      wch[i] = (co < COUT && c0 + c < cin) ? w[((long long)tap * cin + c0 + c) * COUT + co] : 0.f;
    }
    __syncthreads();
    // every lane reads the same weight address (LDS broadcast); channels beyond cin hold zero data and zero weights
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const float* tp = tile + ((ty + tap / 3) * (CO_T + 2) + tx + tap % 3) * CO_STR;
      const float4* wp = reinterpret_cast<const float4*>(wch + tap * CO_CK * 4);
#pragma unroll
      for (int c = 0; c < CO_CK; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(tp + c);
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 w4 = wp[c + q];
          acc[0] = fmaf(vv[q], w4.x, acc[0]);
          if (COUT > 1) acc[1] = fmaf(vv[q], w4.y, acc[1]);
          if (COUT > 2) acc[2] = fmaf(vv[q], w4.z, acc[2]);
          if (COUT > 3) acc[3] = fmaf(vv[q], w4.w, acc[3]);
        }
      }
    }
  }
  const int oy = y0 + ty, ox = x0 + tx;
  if (oy < h && ox < wd) {
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      out[((long long)b * COUT + co) * h * wd + (long long)oy * wd + ox] = acc[co] + (bias ? bias[co] : 0.f);
  }
}

// The same layer for SMALL images (batch 1-2 of the small-batch route): the 16x16-pixel workgroups above give a 32x32 image
// 4 workgroups on 256 CUs (57 us for 8.8 MFLOP, profiles/r02_layers32_b1.txt).  Here a workgroup owns 4x4 output pixels
// (64 workgroups per 32x32 image) and 16 lanes share a pixel, each taking every 16th float4 of the channel axis: inputs
// come straight from global memory (the 6x6 halo of a tile is 23 KB, L2-resident), the 9 cin COUT weights are staged in
// LDS once, GroupNorm+SiLU (optional coef planes) is applied per use, and the 16 partial sums of a pixel are folded
// with four xor-shuffles in a fixed order.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_out_small_kernel(const float* __restrict__ x, const float* __restrict__ coef,
                                                                const float* __restrict__ w, const float* __restrict__ bias,
                                                                float* __restrict__ out, int n, int h, int wd, int cin) {
  extern __shared__ float wl[];                                        // [9][cin][COUT]
  for (int i = threadIdx.x; i < 9 * cin * COUT; i += 256) wl[i] = w[i];
  const int tx_n = (wd + 3) / 4, ty_n = (h + 3) / 4;
  const int b = blockIdx.x / (tx_n * ty_n), t = blockIdx.x % (tx_n * ty_n);
  const int pix = threadIdx.x >> 4, g = threadIdx.x & 15;
  const int oy = (t / tx_n) * 4 + (pix >> 2), ox = (t % tx_n) * 4 + (pix & 3);
  const float* sc = coef ? coef + ((long long)b * 2) * cin : nullptr;
  __syncthreads();
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
  const int c4n = cin >> 2;
  for (int c4 = g; c4 < c4n; c4 += 16) {
    const int c = 4 * c4;
    float4 s4 = make_float4(1.f, 1.f, 1.f, 1.f), t4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (sc) { s4 = *reinterpret_cast<const float4*>(sc + c); t4 = *reinterpret_cast<const float4*>(sc + cin + c); }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int yy = oy + tap / 3 - 1, xx = ox + tap % 3 - 1;
      if (yy < 0 || yy >= h || xx < 0 || xx >= wd || oy >= h || ox >= wd) continue;    // zero padding of the ACTIVATED tensor
      float4 v = *reinterpret_cast<const float4*>(x + (((long long)b * h + yy) * wd + xx) * cin + c);
      if (sc) {
        v.x = silu_f(fmaf(v.x, s4.x, t4.x)); v.y = silu_f(fmaf(v.y, s4.y, t4.y));
        v.z = silu_f(fmaf(v.z, s4.z, t4.z)); v.w = silu_f(fmaf(v.w, s4.w, t4.w));
      }
      const float* wp = wl + ((long long)tap * cin + c) * COUT;
#pragma unroll
      for (int co = 0; co < COUT; ++co)
        acc[co] = fmaf(v.w, wp[3 * COUT + co], fmaf(v.z, wp[2 * COUT + co], fmaf(v.y, wp[COUT + co], fmaf(v.x, wp[co], acc[co]))));
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) acc[co] += __shfl_xor(acc[co], o, 64);
  }
  if (g == 0 && oy < h && ox < wd) {
#pragma unroll
    for (int co = 0; co < COUT; ++co)
      out[((long long)b * COUT + co) * h * wd + (long long)oy * wd + ox] = acc[co] + (bias ? bias[co] : 0.f);
  }
}

__global__ void conv1x1_nchw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                    const float* __restrict__ bias, float* __restrict__ out, int n, int hw, int cin,
                                    int cout) {
  long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)n * hw) return;
  int b = (int)(idx / hw), p = (int)(idx - (long long)b * hw);
  for (int co = 0; co < cout; ++co) {
    float acc = bias ? bias[co] : 0.f;
    for (int c = 0; c < cin; ++c) acc = fmaf(x[((long long)b * cin + c) * hw + p], w[co * cin + c], acc);
    out[((long long)b * cout + co) * hw + p] = acc;
  }
}

// ---------------------------------------------------------------------------------------------
// DDIM update (ddim.py:170-203).  Coefficients come from a device table indexed by a device
// counter, so the same launch replays for every step inside a captured graph.
__global__ void ddim_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ noise, const float* __restrict__ table,
                                 const int* __restrict__ step_idx, float cfg_scale, int cfg, float* __restrict__ x_prev,
                                 float* __restrict__ pred_x0, long long total) {
  const int index = *step_idx;
  const float a_t = table[4 * index], a_prev = table[4 * index + 1], sigma = table[4 * index + 2],
              s1m = table[4 * index + 3];
  // same fp32 operation order as the reference expressions
  const float sqrt_at = sqrtf(a_t);
  const float dir_c = sqrtf(1.0f - a_prev - sigma * sigma);
  const float sqrt_ap = sqrtf(a_prev);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    float e;
    if (cfg) {
      const float eu = eps[i], ec = eps[total + i];
      e = eu + cfg_scale * (ec - eu);
    } else {
      e = eps[i];
    }
    const float xv = x[i];
    const float p0 = (xv - s1m * e) / sqrt_at;
    float xp = sqrt_ap * p0 + dir_c * e;
    if (noise) xp += sigma * noise[i];
    x_prev[i] = xp;
    if (pred_x0) pred_x0[i] = p0;
  }
}

// runs after ddim_step_kernel on the same stream: index -= dir (sampling walks down, inversion up),
// ts[:] = timesteps[index]
__global__ void ddim_advance_kernel(int* step_idx, const long long* __restrict__ timesteps, long long* __restrict__ ts,
                                    int n_ts, int dir, int n_steps) {
  int index = *step_idx - dir;
  if (index < 0) index = 0;
  if (index > n_steps - 1) index = n_steps - 1;
  const long long t = timesteps[index];
  for (int i = threadIdx.x; i < n_ts; i += blockDim.x) ts[i] = t;
  __syncthreads();
  if (threadIdx.x == 0) *step_idx = index;
}

// ancestral DDPM update (ddpm.py:215-228,1049-1109), per-sample timestep
__global__ void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                 const float* __restrict__ noise, const float* __restrict__ tables,
                                 const float* __restrict__ logvar, const long long* __restrict__ t,
                                 float* __restrict__ x_prev, long long per_sample, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int b = (int)(i / per_sample);
    const long long tt = t[b];
    const float c_recip = tables[4 * tt], c_recipm1 = tables[4 * tt + 1], c1 = tables[4 * tt + 2],
                c2 = tables[4 * tt + 3];
    const float xv = x[i];
    const float x0 = c_recip * xv - c_recipm1 * eps[i];
    const float mean = c1 * x0 + c2 * xv;
    const float nz = (tt == 0) ? 0.f : 1.f;
    x_prev[i] = mean + nz * expf(0.5f * logvar[tt]) * (noise ? noise[i] : 0.f);
  }
}

// ---------------------------------------------------------------------------------------------
// VQ nearest codebook entry (quantize.py:276-285).  One wave per latent vector: the 64 lanes split
// the codebook (coalesced reads of e), keep (min d, first index), then a wave-wide arg-min that
// breaks ties towards the smaller index like torch.argmin.
// The distance reproduces the reference's fp32 rounding sequence exactly (checked bit for bit against
// torch-CPU on 6.7e7 distances, tools/make_golden.py --tree vq):
//   |z|^2, |e|^2 : every square rounded, then added left to right   (torch.sum(x ** 2, dim=1))
//   z.e          : k-ordered fused multiply-add chain               (einsum -> sgemm)
//   d            : fl(fl(|z|^2 + |e|^2) - 2 z.e)                    (2 * x is exact)
// so indices are bit-exact, not "equal up to near-ties".  NaN distances order first (torch.argmin
// returns the first NaN), and the winner is always a valid row: a diverged latent never turns into an
// out-of-bounds gather.
__device__ __forceinline__ bool vq_before(float da, int ia, float db, int ib) {
  const bool na = da != da, nb = db != db;
  if (na != nb) return na;
  if (na || da == db) return ia < ib;
  return da < db;
}

template <int DIM>
__global__ __launch_bounds__(256) void vq_nearest_kernel(const float* __restrict__ z, const float* __restrict__ cb,
                                                         float* __restrict__ zq, int* __restrict__ idx_out, int n,
                                                         int hw, int n_embed) {
  const int lane = threadIdx.x & 63;
  const long long pix = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (long long)n * hw) return;
  const int b = (int)(pix / hw), p = (int)(pix - (long long)b * hw);
  float zv[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) zv[d] = z[((long long)b * DIM + d) * hw + p];
  float zz = __fmul_rn(zv[0], zv[0]);
#pragma unroll
  for (int d = 1; d < DIM; ++d) zz = __fadd_rn(zz, __fmul_rn(zv[d], zv[d]));
  float best = INFINITY;
  int besti = lane < n_embed ? lane : 0;       // always a valid row
  for (int j = lane; j < n_embed; j += 64) {
    float e[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) e[d] = cb[(long long)j * DIM + d];
    float ee = __fmul_rn(e[0], e[0]);
    float ze = __fmul_rn(zv[0], e[0]);
#pragma unroll
    for (int d = 1; d < DIM; ++d) {
      ee = __fadd_rn(ee, __fmul_rn(e[d], e[d]));
      ze = __fmaf_rn(zv[d], e[d], ze);
    }
    const float dist = __fsub_rn(__fadd_rn(zz, ee), __fmul_rn(2.0f, ze));
    if (j == lane || vq_before(dist, j, best, besti)) { best = dist; besti = j; }
  }
  int has = lane < n_embed ? 1 : 0;            // lanes beyond a tiny codebook hold no candidate
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ob = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(besti, o, 64);
    const int oh = __shfl_xor(has, o, 64);
    if (oh && (!has || vq_before(ob, oi, best, besti))) { best = ob; besti = oi; has = 1; }
  }
  if (lane < DIM) {
    // the reference returns the straight-through value z + (e - z) (quantize.py:299), which is not e bit for bit
    const float zl = z[((long long)b * DIM + lane) * hw + p];
    zq[((long long)b * DIM + lane) * hw + p] = __fadd_rn(zl, __fsub_rn(cb[(long long)besti * DIM + lane], zl));
  }
  if (lane == 0 && idx_out) idx_out[pix] = besti;
}

// ---------------------------------------------------------------------------------------------
__global__ void permute3_kernel(const float* __restrict__ src, float* __restrict__ dst, int d0, int d1, int d2, int p0,
                                int p1, int p2, long long total) {
  const int sd[3] = {d0, d1, d2};
  const long long ss[3] = {(long long)d1 * d2, d2, 1};
  const int e0 = sd[p0], e1 = sd[p1], e2 = sd[p2];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int i2 = (int)(i % e2);
    long long r = i / e2;
    int i1 = (int)(r % e1);
    int i0 = (int)(r / e1);
    (void)e0;
    dst[i] = src[i0 * ss[p0] + i1 * ss[p1] + i2 * ss[p2]];
  }
}

// Attention heads that are not 32 wide (reference kwargs num_heads / num_head_channels, openaimodel.py:542-549): the flash kernels
// are built for d = 32; other widths go through batched GEMMs on head-major copies.  gather: token rows [n tokens][ld] with the heads
// side by side from column col0 -> [n heads][tokens][dp], the d columns of a head zero-padded to dp (a multiple of 32: exact for
// q k^T, and the padded columns of p v are never read back); scatter: the inverse, first d columns of each head.
__global__ void heads_gather_kernel(const float* __restrict__ src, int ld, int col0, float* __restrict__ dst, int tokens, int heads, int d,
                                    int dp, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % dp);
    long long r = i / dp;
    const int t = (int)(r % tokens);
    r /= tokens;
    const int h = (int)(r % heads);
    const long long b = r / heads;
    dst[i] = j < d ? src[(b * tokens + t) * ld + col0 + h * d + j] : 0.f;
  }
}

__global__ void heads_scatter_kernel(const float* __restrict__ src, float* __restrict__ dst, int ld, int tokens, int heads, int d, int dp,
                                     long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int j = (int)(i % d);
    long long r = i / d;
    const int h = (int)(r % heads);
    r /= heads;
    const int t = (int)(r % tokens);
    const long long b = r / tokens;
    dst[(b * tokens + t) * ld + h * d + j] = src[((b * heads + h) * tokens + t) * dp + j];
  }
}

// conv weight OIHW [O][I][3][3] -> igemm K order [I/32][9][32][O] (channel-chunk major, tap minor)
__global__ void pack_conv3x3_kernel(const float* __restrict__ src, float* __restrict__ dst, int O, int I, long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int o = (int)(i % O);
    long long r = i / O;
    const int kk = (int)(r % 32); r /= 32;
    const int tap = (int)(r % 9);
    const int cc = (int)(r / 9);
    dst[i] = src[((long long)o * I + cc * 32 + kk) * 9 + tap];
  }
}

__global__ void postprocess_frames_kernel(const float* __restrict__ x, float* __restrict__ out, int n, int c, int hw,
                                          long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    int ch = (int)(i % c);
    long long r = i / c;
    int p = (int)(r % hw);
    int b = (int)(r / hw);
    float v = (x[((long long)b * c + ch) * hw + p] + 1.0f) / 2.0f;
    out[i] = fminf(fmaxf(v, 0.f), 1.f);
  }
}

__global__ void add_rowvec_kernel(float* __restrict__ x, const float* __restrict__ vec, int vec_ld, long long rows, int c,
                                  int rps) {
  const long long total4 = rows * (c / 4);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
    long long row = i / (c / 4);
    int c4 = (int)(i - row * (c / 4)) * 4;
    float4 v = *reinterpret_cast<float4*>(x + row * c + c4);
    float4 a = *reinterpret_cast<const float4*>(vec + (row / rps) * vec_ld + c4);
    v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    *reinterpret_cast<float4*>(x + row * c + c4) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// Conv1DTemporalAttention (talking_face/ldm/modules/encoders/modules.py:76-113), one workgroup per sample:
// five Conv1d(k=3, pad=1)+LeakyReLU(0.02) layers 768->192->64->16->4->1 over the T-frame audio window,
// Linear(T,T)+softmax over time, attention-weighted sum of the input features.  Weights packed [3][cin][cout].
constexpr int AA_TMAX = 32;
__global__ __launch_bounds__(256) void audio_attention_kernel(const float* __restrict__ x, int T, int dim,
                                                              const float* const* __restrict__ w,
                                                              const float* const* __restrict__ bias,
                                                              const float* __restrict__ wl, const float* __restrict__ bl,
                                                              float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int chans[6] = {dim, 192, 64, 16, 4, 1};
  float* xin = sm;                               // [(T+2)][dim]  rows 0 and T+1 are the zero padding
  float* bufa = xin + (T + 2) * dim;             // [(T+2)][192]
  float* bufb = bufa + (T + 2) * 192;            // [(T+2)][64]
  float* att = bufb + (T + 2) * 64;              // [T]
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* xb = x + (long long)b * T * dim;
  for (int i = tid; i < (T + 2) * dim; i += 256) {
    const int t = i / dim - 1, c = i - (t + 1) * dim;
    xin[i] = (t >= 0 && t < T) ? xb[(long long)t * dim + c] : 0.f;
  }
  for (int i = tid; i < (T + 2) * 192; i += 256) bufa[i] = 0.f;
  for (int i = tid; i < (T + 2) * 64; i += 256) bufb[i] = 0.f;
  __syncthreads();
  const float* src = xin;
  float* dst = bufa;
  for (int l = 0; l < 5; ++l) {
    const int cin = chans[l], cout = chans[l + 1];
    const float* wl_ = w[l];
    if (tid < cout) {
      float acc[AA_TMAX];
#pragma unroll
      for (int t = 0; t < AA_TMAX; ++t) acc[t] = 0.f;
      for (int k = 0; k < 3; ++k)
        for (int ci = 0; ci < cin; ++ci) {
          const float wv = wl_[((long long)k * cin + ci) * cout + tid];
#pragma unroll
          for (int t = 0; t < AA_TMAX; ++t)
            if (t < T) acc[t] = fmaf(src[(t + k) * cin + ci], wv, acc[t]);
        }
      const float bv = bias[l][tid];
#pragma unroll
      for (int t = 0; t < AA_TMAX; ++t)
        if (t < T) {
          const float v = acc[t] + bv;
          dst[(t + 1) * cout + tid] = v > 0.f ? v : 0.02f * v;
        }
    }
    __syncthreads();
    // next layer reads what this one wrote; ping-pong between the two scratch buffers (padding rows stay zero
    // because every layer is narrower than the previous one and rows 0 / T+1 are never written)
    src = dst;
    dst = (dst == bufa) ? bufb : bufa;
    if (l + 2 <= 5) {
      const int nc = chans[l + 2];
      for (int i = tid; i < (T + 2) * nc; i += 256) {
        const int t = i / nc;
        if (t == 0 || t == T + 1) dst[i] = 0.f;
      }
    }
    __syncthreads();
  }
  // src: [(T+2)][1] conv output; attention = softmax(Linear(T,T)(conv))
  if (tid < T) {
    float a = bl[tid];
    for (int j = 0; j < T; ++j) a = fmaf(wl[tid * T + j], src[j + 1], a);
    att[tid] = a;
  }
  __syncthreads();
  if (tid == 0) {
    float mx = -INFINITY, s = 0.f;
    for (int t = 0; t < T; ++t) mx = fmaxf(mx, att[t]);
    for (int t = 0; t < T; ++t) { att[t] = expf(att[t] - mx); s += att[t]; }
    for (int t = 0; t < T; ++t) att[t] /= s;
  }
  __syncthreads();
  for (int c = tid; c < dim; c += 256) {
    float o = 0.f;
    for (int t = 0; t < T; ++t) o = fmaf(xin[(t + 1) * dim + c], att[t], o);
    out[(long long)b * dim + c] = o;
  }
}

// masked_img[:, y0:, :] = value on NCHW images with a per-image first masked row (MEADBase3, taming/data/custom.py:375-389)
__global__ void mask_rows_kernel(float* __restrict__ img, const int* __restrict__ y0, int c, int h, int w, float value,
                                 long long total) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int y = (int)((i / w) % h);
    const int n = (int)(i / ((long long)c * h * w));
    if (y >= y0[n]) img[i] = value;
  }
}

static inline unsigned grid_for(long long total, int block = 256, int cap = 4096) {
  long long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace ldmk

using namespace ldmk;

// 2.1: folded LayerNorm, Winograd / upsample-phase transforms, split-K up to 64
extern "C" int ldmk_version(void) { return 220; }

namespace ldmk { void igemm_init_attributes(); }

extern "C" int ldmk_init(int device) {
  using namespace ldmk;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    (void)hipGetLastError();
    set_error("ldmk_init: no HIP device (%s)", e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    return LDMK_EHIP;
  }
  LDMK_REQUIRE(device >= 0 && device < count, "ldmk_init: device %d outside [0,%d)", device, count);
  hipDeviceProp_t prop;
  if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess) {
    set_error("ldmk_init: device %d: %s", device, hipGetErrorString(e));
    return LDMK_EHIP;
  }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("ldmk_init: device %d is %s; this library is built for gfx950 (MI355X) only", device, prop.gcnArchName);
    return LDMK_EHIP;
  }
  igemm_init_attributes();
  return check_launch("ldmk_init");
}
extern "C" const char* ldmk_last_error(void) { return g_err; }

extern "C" int ldmk_dense_small(const float* x, int ldx, const float* w, const float* bias, float* out, int ldo,
                                int rows, int K, int N, int silu_in, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && w && out && rows > 0 && K > 0 && N > 0, "ldmk_dense_small: bad args");
  LDMK_REQUIRE(ldx >= K && ldo >= N, "ldmk_dense_small: leading dims");
  using namespace ldmk;
  // 16-byte weight loads whenever the layout allows (every UNet / encoder Linear does)
  // silu_in bit 1 (value 2): the 16-byte-load form for up to 4 batch rows (the batch-1/2 route: 13.7 -> 6.9 us per call).  It is
  // REQUESTED by the caller, never chosen from `rows`: the two forms sum K in different orders, and a sample's result must not
  // depend on how many samples share the launch (a sharded job runs fewer rows per rank than the unsharded one).  With 16 rows
  // per workgroup the form's 64 accumulators per lane and their cross-group sums were SLOWER than the scalar form (33 vs 20 us
  // per call at batch 16) -- measured, not shipped.
  const int want4 = silu_in & 2;
  silu_in &= 1;
  if (want4) {
    LDMK_REQUIRE(rows <= 4 && N % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)out & 15) == 0 &&
                 (!bias || ((uintptr_t)bias & 15) == 0),
                 "ldmk_dense_small: the 16-byte form (silu_in & 2) needs rows <= 4, N and ldo multiples of 4, 16-byte aligned w / bias / out");
    hipLaunchKernelGGL(dense_small4_kernel<4>, dim3((N + 63) / 64, 1), dim3(64 * DS_NW), 0, (hipStream_t)stream, x, ldx, w, bias, out,
                       ldo, rows, K, N, silu_in);
    return check_launch("ldmk_dense_small");
  }
  dim3 grid((N + 63) / 64, (rows + DS_ROWS - 1) / DS_ROWS);
  hipLaunchKernelGGL(dense_small_kernel, grid, dim3(64 * ldmk::DS_NW), 0, (hipStream_t)stream, x, ldx, w, bias, out, ldo, rows, K, N,
                     silu_in);
  return check_launch("ldmk_dense_small");
}

extern "C" int ldmk_timestep_embedding(const long long* t, const float* freqs, float* emb, int n, int dim, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(t && freqs && emb && n > 0 && dim >= 2, "ldmk_timestep_embedding: bad args");
  int total = n * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, t, freqs,
                     emb, n, dim);
  return check_launch("ldmk_timestep_embedding");
}

extern "C" int ldmk_conv3x3_in(const float* x0, int c0, const float* x1, int c1, const float* w, const float* bias,
                               float* out, int n, int h, int w_, int cout, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x0 && w && out && n > 0 && h > 0 && w_ > 0 && cout > 0, "ldmk_conv3x3_in: bad args");
  LDMK_REQUIRE(c0 > 0 && c0 + c1 <= CIN_MAX && (c1 == 0) == (x1 == nullptr), "ldmk_conv3x3_in: c0+c1 must be <= 16");
  const int segs = (w_ + CIN_SEG - 1) / CIN_SEG;
  const long long blocks = (long long)n * h * segs;
  LDMK_REQUIRE(blocks < (1LL << 31), "ldmk_conv3x3_in: too many row segments");
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
#define LDMK_CIN_CASE(C) case C: hipLaunchKernelGGL(conv3x3_in_kernel<C>, grid, block, 0, st, x0, c0, x1, c1, w, bias, out, n, h, w_, cout); break;
  switch (c0 + c1) {
    LDMK_CIN_CASE(1) LDMK_CIN_CASE(2) LDMK_CIN_CASE(3) LDMK_CIN_CASE(4) LDMK_CIN_CASE(5) LDMK_CIN_CASE(6) LDMK_CIN_CASE(7) LDMK_CIN_CASE(8)
    LDMK_CIN_CASE(9) LDMK_CIN_CASE(10) LDMK_CIN_CASE(11) LDMK_CIN_CASE(12) LDMK_CIN_CASE(13) LDMK_CIN_CASE(14) LDMK_CIN_CASE(15) LDMK_CIN_CASE(16)
  }
#undef LDMK_CIN_CASE
  return check_launch("ldmk_conv3x3_in");
}

extern "C" int ldmk_conv3x3_out(const float* x, const float* coef, const float* w, const float* bias, float* out, int n,
                                int h, int w_, int cin, int cout, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && w && out && n > 0 && h > 0 && w_ > 0 && cin > 0, "ldmk_conv3x3_out: bad args");
  LDMK_REQUIRE(cout >= 1 && cout <= 4, "ldmk_conv3x3_out: cout=%d must be in [1,4]", cout);
  LDMK_REQUIRE(cin % 4 == 0, "ldmk_conv3x3_out: cin=%d must be a multiple of 4", cin);
  const long long blocks = (long long)n * ((h + CO_T - 1) / CO_T) * ((w_ + CO_T - 1) / CO_T);
  LDMK_REQUIRE(blocks < (1LL << 31), "ldmk_conv3x3_out: too many tiles");
  const dim3 grid((unsigned)blocks), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (cout) {
    case 1: hipLaunchKernelGGL(conv3x3_out_kernel<1>, grid, block, 0, st, x, coef, w, bias, out, n, h, w_, cin); break;
    case 2: hipLaunchKernelGGL(conv3x3_out_kernel<2>, grid, block, 0, st, x, coef, w, bias, out, n, h, w_, cin); break;
    case 3: hipLaunchKernelGGL(conv3x3_out_kernel<3>, grid, block, 0, st, x, coef, w, bias, out, n, h, w_, cin); break;
    default: hipLaunchKernelGGL(conv3x3_out_kernel<4>, grid, block, 0, st, x, coef, w, bias, out, n, h, w_, cin); break;
  }
  return check_launch("ldmk_conv3x3_out");
}

extern "C" int ldmk_conv3x3_out_small(const float* x, const float* coef, const float* w, const float* bias, float* out, int n,
                                      int h, int w_, int cin, int cout, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(x && w && out && n > 0 && h > 0 && w_ > 0 && cin > 0, "ldmk_conv3x3_out_small: bad args");
  LDMK_REQUIRE(cout >= 1 && cout <= 4, "ldmk_conv3x3_out_small: cout=%d must be in [1,4]", cout);
  LDMK_REQUIRE(cin % 4 == 0 && 9 * cin * cout * 4 <= 60 * 1024, "ldmk_conv3x3_out_small: cin=%d must be a multiple of 4 and the "
               "weights must fit 60 KB of LDS", cin);
  const long long blocks = (long long)n * ((h + 3) / 4) * ((w_ + 3) / 4);
  LDMK_REQUIRE(blocks < (1LL << 31), "ldmk_conv3x3_out_small: too many tiles");
  const dim3 grid((unsigned)blocks), block(256);
  const size_t lds = (size_t)9 * cin * cout * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  switch (cout) {
    case 1: hipLaunchKernelGGL(conv3x3_out_small_kernel<1>, grid, block, lds, st, x, coef, w, bias, out, n, h, w_, cin); break;
    case 2: hipLaunchKernelGGL(conv3x3_out_small_kernel<2>, grid, block, lds, st, x, coef, w, bias, out, n, h, w_, cin); break;
    case 3: hipLaunchKernelGGL(conv3x3_out_small_kernel<3>, grid, block, lds, st, x, coef, w, bias, out, n, h, w_, cin); break;
    default: hipLaunchKernelGGL(conv3x3_out_small_kernel<4>, grid, block, lds, st, x, coef, w, bias, out, n, h, w_, cin); break;
  }
  return check_launch("ldmk_conv3x3_out_small");
}

extern "C" int ldmk_conv1x1_nchw(const float* x, const float* w, const float* bias, float* out, int n, int hw, int cin,
                                 int cout, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && w && out && n > 0 && hw > 0 && cin > 0 && cout > 0, "ldmk_conv1x1_nchw: bad args");
  long long total = (long long)n * hw;
  hipLaunchKernelGGL(conv1x1_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, w,
                     bias, out, n, hw, cin, cout);
  return check_launch("ldmk_conv1x1_nchw");
}

extern "C" int ldmk_ddim_step(const float* x, const float* eps, const float* noise, const float* table,
                              int* step_idx, float cfg_scale, int cfg, float* x_prev, float* pred_x0,
                              long long per_sample, int n, const long long* timesteps, long long* ts, int n_ts,
                              int advance, int n_steps, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && eps && table && step_idx && x_prev && per_sample > 0 && n > 0, "ldmk_ddim_step: bad args");
  if (advance) LDMK_REQUIRE(timesteps && ts && n_ts > 0 && n_steps > 0 && (advance == 1 || advance == -1),
                            "ldmk_ddim_step: advance (+1 down / -1 up) needs timesteps/ts/n_steps");
  long long total = per_sample * n;
  hipLaunchKernelGGL(ddim_step_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, eps, noise, table,
                     step_idx, cfg_scale, cfg, x_prev, pred_x0, total);
  if (advance)
    hipLaunchKernelGGL(ddim_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_idx,
                       timesteps, ts, n_ts, advance, n_steps);
  return check_launch("ldmk_ddim_step");
}

extern "C" int ldmk_advance_timestep(int* step_idx, const long long* timesteps, long long* ts, int n_ts, int advance,
                                     int n_steps, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(step_idx && timesteps && ts && n_ts > 0 && n_steps > 0 && (advance == 1 || advance == -1),
               "ldmk_advance_timestep: bad args (advance is +1 down / -1 up)");
  hipLaunchKernelGGL(ddim_advance_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, step_idx, timesteps, ts, n_ts, advance,
                     n_steps);
  return check_launch("ldmk_advance_timestep");
}

extern "C" int ldmk_ddpm_step(const float* x, const float* eps, const float* noise, const float* tables,
                              const float* logvar, const long long* t, float* x_prev, long long per_sample, int n,
                              void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && eps && tables && logvar && t && x_prev && per_sample > 0 && n > 0, "ldmk_ddpm_step: bad args");
  long long total = per_sample * n;
  hipLaunchKernelGGL(ddpm_step_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, eps, noise, tables,
                     logvar, t, x_prev, per_sample, total);
  return check_launch("ldmk_ddpm_step");
}

extern "C" int ldmk_vq_nearest(const float* z, const float* codebook, float* zq, int* idx, int n, int hw, int dim,
                               int n_embed, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(z && codebook && zq && n > 0 && hw > 0 && n_embed > 0, "ldmk_vq_nearest: bad args");
  long long total = (long long)n * hw;
  dim3 grid((unsigned)((total + 3) / 4));
  hipStream_t st = (hipStream_t)stream;
  switch (dim) {
    case 3: hipLaunchKernelGGL(vq_nearest_kernel<3>, grid, dim3(256), 0, st, z, codebook, zq, idx, n, hw, n_embed); break;
    case 4: hipLaunchKernelGGL(vq_nearest_kernel<4>, grid, dim3(256), 0, st, z, codebook, zq, idx, n, hw, n_embed); break;
    default: LDMK_REQUIRE(false, "ldmk_vq_nearest: embed dim %d unsupported (3 or 4)", dim);
  }
  return check_launch("ldmk_vq_nearest");
}

extern "C" int ldmk_heads_gather(const float* src, int ld, int col0, float* dst, int n, int tokens, int heads, int d, int dp, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(src && dst && n > 0 && tokens > 0 && heads > 0 && d > 0 && dp >= d && col0 >= 0 && ld >= col0 + heads * d,
               "ldmk_heads_gather: bad args (ld=%d col0=%d heads=%d d=%d dp=%d)", ld, col0, heads, d, dp);
  const long long total = (long long)n * heads * tokens * dp;
  hipLaunchKernelGGL(heads_gather_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, ld, col0, dst, tokens, heads, d, dp, total);
  return check_launch("ldmk_heads_gather");
}

extern "C" int ldmk_heads_scatter(const float* src, float* dst, int ld, int n, int tokens, int heads, int d, int dp, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(src && dst && n > 0 && tokens > 0 && heads > 0 && d > 0 && dp >= d && ld >= heads * d,
               "ldmk_heads_scatter: bad args (ld=%d heads=%d d=%d dp=%d)", ld, heads, d, dp);
  const long long total = (long long)n * tokens * heads * d;
  hipLaunchKernelGGL(heads_scatter_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, ld, tokens, heads, d, dp, total);
  return check_launch("ldmk_heads_scatter");
}

extern "C" int ldmk_permute3(const float* src, float* dst, int d0, int d1, int d2, int p0, int p1, int p2, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(src && dst && d0 > 0 && d1 > 0 && d2 > 0, "ldmk_permute3: bad args");
  LDMK_REQUIRE(((1 << p0) | (1 << p1) | (1 << p2)) == 7, "ldmk_permute3: perm must be a permutation of 0,1,2");
  long long total = (long long)d0 * d1 * d2;
  hipLaunchKernelGGL(permute3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, d0, d1, d2, p0,
                     p1, p2, total);
  return check_launch("ldmk_permute3");
}

extern "C" int ldmk_audio_attention(const float* x, int n, int T, int dim, const float* const* conv_w,
                                    const float* const* conv_b, const float* lin_w, const float* lin_b, float* out,
                                    void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && conv_w && conv_b && lin_w && lin_b && out && n > 0 && dim > 0, "ldmk_audio_attention: bad args");
  LDMK_REQUIRE(T >= 1 && T <= AA_TMAX, "ldmk_audio_attention: window T=%d outside [1,%d]", T, AA_TMAX);
  size_t lds = ((size_t)(T + 2) * (dim + 192 + 64) + T) * sizeof(float);
  LDMK_REQUIRE(lds <= 160 * 1024, "ldmk_audio_attention: window too large for LDS");
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(audio_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(audio_attention_kernel, dim3(n), dim3(256), lds, (hipStream_t)stream, x, T, dim, conv_w, conv_b, lin_w,
                     lin_b, out);
  return check_launch("ldmk_audio_attention");
}

extern "C" int ldmk_mask_rows(float* img, const int* y0, int n, int c, int h, int w, float value, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(img && y0 && n > 0 && c > 0 && h > 0 && w > 0, "ldmk_mask_rows: bad args");
  long long total = (long long)n * c * h * w;
  hipLaunchKernelGGL(mask_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, img, y0, c, h, w, value, total);
  return check_launch("ldmk_mask_rows");
}

extern "C" int ldmk_pack_conv3x3(const float* src, float* dst, int cout, int cin, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(src && dst && cout > 0 && cin > 0 && cin % 32 == 0, "ldmk_pack_conv3x3: cin=%d must be a multiple of 32", cin);
  long long total = (long long)cout * cin * 9;
  hipLaunchKernelGGL(pack_conv3x3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, cout, cin, total);
  return check_launch("ldmk_pack_conv3x3");
}

extern "C" int ldmk_postprocess_frames(const float* x, float* out, int n, int c, int hw, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && out && n > 0 && c > 0 && hw > 0, "ldmk_postprocess_frames: bad args");
  long long total = (long long)n * c * hw;
  hipLaunchKernelGGL(postprocess_frames_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, out, n, c,
                     hw, total);
  return check_launch("ldmk_postprocess_frames");
}

extern "C" int ldmk_add_rowvec(float* x, const float* vec, int vec_ld, long long rows, int c, int rows_per_sample,
                               void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(x && vec && rows > 0 && c > 0 && c % 4 == 0 && vec_ld % 4 == 0 && rows_per_sample > 0,
               "ldmk_add_rowvec: bad args");
  hipLaunchKernelGGL(add_rowvec_kernel, dim3(grid_for(rows * (c / 4))), dim3(256), 0, (hipStream_t)stream, x, vec, vec_ld,
                     rows, c, rows_per_sample);
  return check_launch("ldmk_add_rowvec");
}
