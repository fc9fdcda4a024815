// Weight-gradient GEMM on the f32 matrix cores: dW[Kw][N] = sum over rows r of A[r][kw] * dY[r][n].
//
// The transposed-A counterpart of igemm.hip for the training step (ddpm.py:1014-1047 p_losses,
// backward of every nn.Conv2d 3x3 / 1x1 and nn.Linear of the UNet).  The reduction runs over the n*H*W token
// rows, so both operands are staged as they lie in HBM -- a 32-row slice of A (rows mode: the saved GEMM input;
// conv mode: the im2col gather of the saved NHWC input, same index arithmetic as the forward gather including
// stride, asymmetric pad and nearest-x2 upsampling) and the matching 32 rows of dY -- no transposes anywhere.
// dW comes out in the packed layout the forward consumes ([Cin/32][9][32][Cout] rows for 3x3 convolutions), so
// the optimizer updates packed weights in place.  The output is small and the reduction long: the rows are split
// over blockIdx.y (`splitr`) into partial slabs that a second kernel sums in a fixed order (bitwise reproducible).
#include "ldmk_common.h"
#include <stdlib.h>

namespace ldmk {

typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));

// BF = true: bf16 matrix-core compute (BASELINE configs[4]).  Staging is the fp32 kernel's, unchanged (float4 loads, one
// index computation per 16 bytes, r-major fp32 LDS slices).  The MFMA's K dimension is the ROW index r here, so a bf16
// fragment (8 consecutive r of one column) is gathered from the slice with 8 strided ds_read_b32 -- consecutive lanes read
// consecutive columns: conflict-free -- and rounded to bf16 (RNE) in registers.  2 TM TN MFMAs (v_mfma_f32_32x32x16_bf16)
// per 32-row slice instead of 16 TM TN fp32 ones; accumulation, slabs and the bias column sums stay fp32.  (First
// version: a transposed bf16 LDS image filled by 4 dword loads per thread and element-wise conv index arithmetic --
// 40 % SLOWER than the fp32 kernel, the staging instructions cost more than the matrix work saved.)
template <int TM, int TN, int WM, int WN, bool BF = false>
__global__ __launch_bounds__(256) void wgrad_kernel(const ldmk_wgrad_args p, const int splitr, float* __restrict__ ws) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int ASTR = BM + 4, BSTR = BN + 4;       // 16-B aligned rows for the float4 staging stores
  constexpr int NA = BM / 32, NB = BN / 32;         // float4 per thread per 32-row slice
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) float As[32 * ASTR];
  __shared__ __attribute__((aligned(16))) float Bs[32 * BSTR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  const int tiles_m = (p.Kw + BM - 1) / BM;
  const int m0 = (blockIdx.x % tiles_m) * BM, n0 = (blockIdx.x / tiles_m) * BN;
  const int ks = blockIdx.y, bz = blockIdx.z;
  const float* __restrict__ ap = p.a + (long long)bz * p.a_bstride;
  const float* __restrict__ dyp = p.dy + (long long)bz * p.dy_bstride;
  const bool conv = p.a_mode == LDMK_A_CONV3X3;
  const int rps = p.out_h * p.out_w;
  // power-of-two feature maps (every UNet level): row -> (sample, y, x) by shifts instead of integer divisions
  const int rps_shift = (rps & (rps - 1)) == 0 ? __ffs(rps) - 1 : -1;
  const int ow_shift = (p.out_w & (p.out_w - 1)) == 0 ? __ffs(p.out_w) - 1 : -1;

  const int iters_all = (p.R + 31) / 32;
  const int it_per = (iters_all + splitr - 1) / splitr;
  const int it_begin = ks * it_per, it_end = min(iters_all, it_begin + it_per);

  // per-thread, loop-invariant column bookkeeping of the A slice
  int a_rl[NA], a_col[NA], a_ch[NA], a_dy[NA], a_dx[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + 256 * i;
    a_rl[i] = idx / (BM / 4);
    a_col[i] = (idx - a_rl[i] * (BM / 4)) * 4;
    const int kw = m0 + a_col[i];
    a_ok[i] = kw < p.Kw;
    if (conv) {
      const int chunk = kw >> 5, cc = chunk / 9, tap = chunk - cc * 9;
      a_ch[i] = cc * 32 + (kw & 31);
      a_dy[i] = tap / 3;
      a_dx[i] = tap - a_dy[i] * 3;
    } else {
      a_ch[i] = kw; a_dy[i] = 0; a_dx[i] = 0;
    }
  }
  int b_rl[NB], b_col[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + 256 * i;
    b_rl[i] = idx / (BN / 4);
    b_col[i] = (idx - b_rl[i] * (BN / 4)) * 4;
  }

  float4 areg[NA], breg[NB];
  // one element of the (virtual) A matrix: row r of the reduction, column described by (ch, dy, dx)
  auto a_elem_ptr = [&](int r, int ch, int ddy, int ddx) -> const float* {
    if (!conv) return ap + (long long)r * p.lda + ch;
    int n, pix, oy, ox;
    if (rps_shift >= 0) { n = r >> rps_shift; pix = r & (rps - 1); } else { n = r / rps; pix = r - n * rps; }
    if (ow_shift >= 0) { oy = pix >> ow_shift; ox = pix & (p.out_w - 1); } else { oy = pix / p.out_w; ox = pix - oy * p.out_w; }
    int iy = oy * p.stride - p.pad_lo + ddy, ix = ox * p.stride - p.pad_lo + ddx;
    const int lim_h = p.upsample ? 2 * p.in_h : p.in_h, lim_w = p.upsample ? 2 * p.in_w : p.in_w;
    if (iy < 0 || ix < 0 || iy >= lim_h || ix >= lim_w) return nullptr;
    if (p.upsample) { iy >>= 1; ix >>= 1; }
    return ap + ((long long)(n * p.in_h + iy) * p.in_w + ix) * p.c + ch;
  };
  auto load_slice = [&](int it) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = it * 32 + a_rl[i];
      if (a_ok[i] && r < p.R) {
        const float* src = a_elem_ptr(r, a_ch[i], a_dy[i], a_dx[i]);
        if (src) v = *reinterpret_cast<const float4*>(src);
      }
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = it * 32 + b_rl[i], n = n0 + b_col[i];
      if (r < p.R && n < p.N) v = *reinterpret_cast<const float4*>(dyp + (long long)r * p.ldy + n);
      breg[i] = v;
    }
  };
  auto store_slice = [&]() {
#pragma unroll
    for (int i = 0; i < NA; ++i) *reinterpret_cast<float4*>(As + a_rl[i] * ASTR + a_col[i]) = areg[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(Bs + b_rl[i] * BSTR + b_col[i]) = breg[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* Aw = As + half * ASTR + wm * (32 * TM) + l31;
  const float* Bw = Bs + half * BSTR + wn * (32 * TN) + l31;
  // bias gradient = column sums of dY: the workgroups of the first row tile add up the dY slices they stage anyway
  const bool do_bias = p.dbias != nullptr && m0 == 0 && tid < BN;
  float bsum = 0.f;
  if (it_begin < it_end) load_slice(it_begin);
  for (int it = it_begin; it < it_end; ++it) {
    __syncthreads();
    store_slice();
    __syncthreads();
    if (it + 1 < it_end) load_slice(it + 1);
    if (do_bias) {
#pragma unroll
      for (int r = 0; r < 32; ++r) bsum += Bs[r * BSTR + tid];
    }
    if constexpr (BF) {
      // fragment of k-step s: rows 16 s + 8 half + e (e = 0..7) of this lane's column, rounded to bf16
      const float* Ag = As + (8 * half) * ASTR + wm * (32 * TM) + l31;
      const float* Bg = Bs + (8 * half) * BSTR + wn * (32 * TN) + l31;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        wbf16x8 a8[TM], b8[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e) a8[i][e] = (__bf16)Ag[(16 * s + e) * ASTR + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) b8[j][e] = (__bf16)Bg[(16 * s + e) * BSTR + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
      }
      continue;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = Aw[2 * s * ASTR + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bw[2 * s * BSTR + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int rowbase = m0 + wm * (32 * TM), colbase = n0 + wn * (32 * TN);
  float* dst;
  long long ld;
  const int srows = p.Kw + (p.dbias ? 1 : 0);      // slab rows: dW plus one row of bias partial sums
  if (do_bias && n0 + tid < p.N) {
    if (splitr > 1) ws[(((long long)bz * splitr + ks) * srows + p.Kw) * p.N + n0 + tid] = bsum;
    else p.dbias[n0 + tid] = p.accumulate ? p.dbias[n0 + tid] + p.alpha * bsum : p.alpha * bsum;
  }
  if (splitr > 1) {
    dst = ws + ((long long)bz * splitr + ks) * srows * p.N;
    ld = p.N;
  } else {
    dst = p.dw + (long long)bz * p.dw_bstride;
    ld = p.ldw;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colbase + j * 32 + l31;
    if (col >= p.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < p.Kw) {
          float* d = dst + (long long)row * ld + col;
          if (splitr > 1) *d = acc[i][j][r];
          else *d = p.accumulate ? *d + p.alpha * acc[i][j][r] : p.alpha * acc[i][j][r];
        }
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------------------
// The bf16 weight-gradient GEMM with HARDWARE-TRANSPOSED fragment reads (round 5; BASELINE configs[4]).  The MFMA's K index is the
// token ROW r, which is the slow index of both operands as they lie in HBM ([r][kw] and [r][n] rows).  wgrad_kernel<BF = true>
// staged fp32 slices and gathered every bf16 fragment with eight strided ds_read_b32 + eight conversions: 96 LDS reads and 96
// v_cvt per ten MFMAs on the 128x160 tile -- the matrix pipe idled behind them (62 us per launch, 10 ms of the 60 ms step).
// Here the slices are rounded to bf16 ONCE while they are staged (row-major [r][col] images, 8-byte stores) and gfx950's
// ds_read_b64_tr_b16 delivers them column-major: per 16-lane group a block of 4 rows x 16 columns, lane i receiving column i --
// two reads make the 8 consecutive r of one column a 32x32x16 operand wants (lane = column, half-wave = which 8 rows).  Row
// strides of 64 x odd bytes keep the four rows of a half-wave's read on disjoint banks.  Two LDS buffers, one barrier per
// 32-row slice, the next slice's global loads in flight under the MFMAs.  Same split over blockIdx.y, same fixed-order slab
// reduce, same output layout as wgrad_kernel; the bias gradient (column sums of dY) is accumulated in fp32 from the registers
// the slices pass through and combined once at the end in a fixed order.
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));
constexpr int wg_tr_stride(int cols) {             // bf16 elements per LDS row: the smallest 64 x odd bytes >= 2 cols
  int k = (cols * 2 + 63) / 64;
  if (k % 2 == 0) ++k;
  return k * 32;
}

template <int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(256) void wgrad_tr_kernel(const ldmk_wgrad_args p, const int splitr, float* __restrict__ ws) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  constexpr int AS = wg_tr_stride(BM), BS = wg_tr_stride(BN);
  constexpr int NA = BM / 32, NB = BN / 32;         // float4 per thread per 32-row slice
  static_assert(WM * WN == 4, "4 waves per workgroup");
  __shared__ __attribute__((aligned(16))) __bf16 At[2][32 * AS];
  __shared__ __attribute__((aligned(16))) __bf16 Bt[2][32 * BS];
  __shared__ __attribute__((aligned(16))) float4 bpart[NB * 256];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, l31 = lane & 31, half = lane >> 5;
  const int tiles_m = (p.Kw + BM - 1) / BM;
  const int m0 = (blockIdx.x % tiles_m) * BM, n0 = (blockIdx.x / tiles_m) * BN;
  const int ks = blockIdx.y, bz = blockIdx.z;
  const float* __restrict__ ap = p.a + (long long)bz * p.a_bstride;
  const float* __restrict__ dyp = p.dy + (long long)bz * p.dy_bstride;
  const bool conv = p.a_mode == LDMK_A_CONV3X3;
  const int rps = p.out_h * p.out_w;
  const int rps_shift = (rps & (rps - 1)) == 0 ? __ffs(rps) - 1 : -1;
  const int ow_shift = (p.out_w & (p.out_w - 1)) == 0 ? __ffs(p.out_w) - 1 : -1;
  const int iters_all = (p.R + 31) / 32;
  const int it_per = (iters_all + splitr - 1) / splitr;
  const int it_begin = ks * it_per, it_end = min(iters_all, it_begin + it_per);

  int a_rl[NA], a_col[NA], a_ch[NA], a_dy[NA], a_dx[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int idx = tid + 256 * i;
    a_rl[i] = idx / (BM / 4);
    a_col[i] = (idx - a_rl[i] * (BM / 4)) * 4;
    const int kw = m0 + a_col[i];
    a_ok[i] = kw < p.Kw;
    if (conv) {
      const int chunk = kw >> 5, cc = chunk / 9, tap = chunk - cc * 9;
      a_ch[i] = cc * 32 + (kw & 31);
      a_dy[i] = tap / 3;
      a_dx[i] = tap - a_dy[i] * 3;
    } else {
      a_ch[i] = kw; a_dy[i] = 0; a_dx[i] = 0;
    }
  }
  int b_rl[NB], b_col[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int idx = tid + 256 * i;
    b_rl[i] = idx / (BN / 4);
    b_col[i] = (idx - b_rl[i] * (BN / 4)) * 4;
  }
  float4 areg[NA], breg[NB], bsum[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) bsum[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  auto a_elem_ptr = [&](int r, int ch, int ddy, int ddx) -> const float* {
    if (!conv) return ap + (long long)r * p.lda + ch;
    int n, pix, oy, ox;
    if (rps_shift >= 0) { n = r >> rps_shift; pix = r & (rps - 1); } else { n = r / rps; pix = r - n * rps; }
    if (ow_shift >= 0) { oy = pix >> ow_shift; ox = pix & (p.out_w - 1); } else { oy = pix / p.out_w; ox = pix - oy * p.out_w; }
    int iy = oy * p.stride - p.pad_lo + ddy, ix = ox * p.stride - p.pad_lo + ddx;
    const int lim_h = p.upsample ? 2 * p.in_h : p.in_h, lim_w = p.upsample ? 2 * p.in_w : p.in_w;
    if (iy < 0 || ix < 0 || iy >= lim_h || ix >= lim_w) return nullptr;
    if (p.upsample) { iy >>= 1; ix >>= 1; }
    return ap + ((long long)(n * p.in_h + iy) * p.in_w + ix) * p.c + ch;
  };
  auto load_slice = [&](int it) {
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = it * 32 + a_rl[i];
      if (a_ok[i] && r < p.R) {
        const float* src = a_elem_ptr(r, a_ch[i], a_dy[i], a_dx[i]);
        if (src) v = *reinterpret_cast<const float4*>(src);
      }
      areg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      const int r = it * 32 + b_rl[i], n = n0 + b_col[i];
      if (r < p.R && n < p.N) v = *reinterpret_cast<const float4*>(dyp + (long long)r * p.ldy + n);
      breg[i] = v;
    }
  };
  auto store_slice = [&](int buf) {                 // rounded to bf16 (RNE) once, here
#pragma unroll
    for (int i = 0; i < NA; ++i)
      *reinterpret_cast<wbf16x4*>(&At[buf][a_rl[i] * AS + a_col[i]]) = wbf16x4{(__bf16)areg[i].x, (__bf16)areg[i].y, (__bf16)areg[i].z, (__bf16)areg[i].w};
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      *reinterpret_cast<wbf16x4*>(&Bt[buf][b_rl[i] * BS + b_col[i]]) = wbf16x4{(__bf16)breg[i].x, (__bf16)breg[i].y, (__bf16)breg[i].z, (__bf16)breg[i].w};
      bsum[i].x += breg[i].x; bsum[i].y += breg[i].y; bsum[i].z += breg[i].z; bsum[i].w += breg[i].w;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // transposed-read addressing: 16-lane group g = lane / 16 reads the block of rows 8 (g / 2) + {0..3} (+ 4 for the second
  // read), columns 16 (g % 2) + {0..15} of an operand tile; lane 4 q + p of the group supplies row q, columns 4 p .. 4 p + 3
  const int g16 = lane >> 4, j16 = lane & 15;
  const int tr_row = 8 * (g16 >> 1) + (j16 >> 2), tr_col = 16 * (g16 & 1) + 4 * (j16 & 3);
  const int a_off = tr_row * AS + wm * (32 * TM) + tr_col;
  const int b_off = tr_row * BS + wn * (32 * TN) + tr_col;
  auto tr8 = [](const __bf16* q, int stride4) -> wbf16x8 {       // rows +0..3 and +4..7 of this lane's column
    const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ws16x4 __attribute__((address_space(3)))*)(q));
    const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((ws16x4 __attribute__((address_space(3)))*)(q + stride4));
    return __builtin_bit_cast(wbf16x8, ws16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
  };

  if (it_begin < it_end) {
    load_slice(it_begin);
    store_slice(0);
  }
  __syncthreads();
  for (int it = it_begin; it < it_end; ++it) {
    const int buf = (it - it_begin) & 1;
    const bool more = it + 1 < it_end;
    if (more) load_slice(it + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      wbf16x8 a8[TM], b8[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a8[i] = tr8(&At[buf][a_off + 16 * s * AS + 32 * i], 4 * AS);
#pragma unroll
      for (int j = 0; j < TN; ++j) b8[j] = tr8(&Bt[buf][b_off + 16 * s * BS + 32 * j], 4 * BS);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
    }
    if (more) store_slice(buf ^ 1);
    __syncthreads();
  }

  const int rowbase = m0 + wm * (32 * TM), colbase = n0 + wn * (32 * TN);
  const int srows = p.Kw + (p.dbias ? 1 : 0);
  if (p.dbias != nullptr && m0 == 0) {              // (workgroup-uniform) bias gradient: column sums of dY, fixed order
#pragma unroll
    for (int i = 0; i < NB; ++i) bpart[i * 256 + tid] = bsum[i];
    __syncthreads();
    if (tid < BN && n0 + tid < p.N) {
      float t = 0.f;
      for (int rl = 0; rl < 32; ++rl) {
        const int idx = rl * (BN / 4) + (tid >> 2);
        t += reinterpret_cast<const float*>(&bpart[(idx >> 8) * 256 + (idx & 255)])[tid & 3];
      }
      if (splitr > 1) ws[(((long long)bz * splitr + ks) * srows + p.Kw) * p.N + n0 + tid] = t;
      else p.dbias[n0 + tid] = p.accumulate ? p.dbias[n0 + tid] + p.alpha * t : p.alpha * t;
    }
  }
  float* dst;
  long long ld;
  if (splitr > 1) {
    dst = ws + ((long long)bz * splitr + ks) * srows * p.N;
    ld = p.N;
  } else {
    dst = p.dw + (long long)bz * p.dw_bstride;
    ld = p.ldw;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colbase + j * 32 + l31;
    if (col >= p.N) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < p.Kw) {
          float* d = dst + (long long)row * ld + col;
          if (splitr > 1) *d = acc[i][j][r];
          else *d = p.accumulate ? *d + p.alpha * acc[i][j][r] : p.alpha * acc[i][j][r];
        }
      }
  }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const ldmk_wgrad_args p, const int splitr,
                                                           const float* __restrict__ ws) {
  const int n4 = p.N / 4;
  const int srows = p.Kw + (p.dbias ? 1 : 0);
  const long long total = (long long)srows * n4;
  const int bz = blockIdx.z;
  const float* slab0 = ws + (long long)bz * splitr * srows * p.N;
  float* outp = p.dw + (long long)bz * p.dw_bstride;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / n4), col = (int)(i - (long long)row * n4) * 4;
    // eight slabs requested at a time, summed in slab order (the split runs to dozens of slabs: one load per trip, each waiting
    // for itself, made this kernel a chain of memory round trips)
    const float* sp = slab0 + (long long)row * p.N + col;
    const long long sstride = (long long)srows * p.N;
    float4 s = *reinterpret_cast<const float4*>(sp);
    for (int k0 = 1; k0 < splitr; k0 += 8) {
      float4 t[8];
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) t[kk] = *reinterpret_cast<const float4*>(sp + (long long)min(k0 + kk, splitr - 1) * sstride);
#pragma unroll
      for (int kk = 0; kk < 8; ++kk)
        if (k0 + kk < splitr) { s.x += t[kk].x; s.y += t[kk].y; s.z += t[kk].z; s.w += t[kk].w; }
    }
    s.x *= p.alpha; s.y *= p.alpha; s.z *= p.alpha; s.w *= p.alpha;
    float* d = row < p.Kw ? outp + (long long)row * p.ldw + col : p.dbias + col;
    if (p.accumulate) { s.x += d[0]; s.y += d[1]; s.z += d[2]; s.w += d[3]; }
    d[0] = s.x; d[1] = s.y; d[2] = s.z; d[3] = s.w;
  }
}

template <int TM, int TN, int WM, int WN, bool BF>
static int launch_wgrad_t(const ldmk_wgrad_args& a, int splitr, hipStream_t st);

template <int TM, int TN, int WM, int WN>
static int launch_wgrad(const ldmk_wgrad_args& a, int splitr, hipStream_t st) {
  return a.compute == LDMK_COMPUTE_BF16 ? launch_wgrad_t<TM, TN, WM, WN, true>(a, splitr, st)
                                        : launch_wgrad_t<TM, TN, WM, WN, false>(a, splitr, st);
}

template <int TM, int TN, int WM, int WN, bool BF>
static int launch_wgrad_t(const ldmk_wgrad_args& a, int splitr, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  const int tiles = ((a.Kw + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  const int nb = a.batch > 1 ? a.batch : 1;
  static const bool tr = [] { const char* e = getenv("LDMK_WGRAD_TR"); return !e || atoi(e) != 0; }();      // (=0: the round-2 gather, A/B only)
  if (BF && tr)
    hipLaunchKernelGGL((wgrad_tr_kernel<TM, TN, WM, WN>), dim3(tiles, splitr, nb), dim3(256), 0, st, a, splitr, a.ws);
  else
    hipLaunchKernelGGL((wgrad_kernel<TM, TN, WM, WN, BF>), dim3(tiles, splitr, nb), dim3(256), 0, st, a, splitr, a.ws);
  if (splitr > 1) {
    long long total = (long long)(a.Kw + (a.dbias ? 1 : 0)) * (a.N / 4);
    int g = (int)((total + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(g, 1, nb), dim3(256), 0, st, a, splitr, a.ws);
  }
  return check_launch("ldmk_wgrad");
}

// tile orientation: put the 160-wide side on whichever of (Kw, N) is a multiple of 160
static int wgrad_cfg(const ldmk_wgrad_args& a) {
  auto waste = [](int v, int b) { return (float)(((v + b - 1) / b) * b) / (float)v; };
  const float w1 = waste(a.Kw, 128) * waste(a.N, 160), w2 = waste(a.Kw, 160) * waste(a.N, 128),
              w3 = waste(a.Kw, 128) * waste(a.N, 128) * 1.08f, w4 = waste(a.Kw, 128) * waste(a.N, 32) * 1.5f;
  int best = 1;
  float bw = w1;
  if (w2 < bw) { bw = w2; best = 2; }
  if (w3 < bw) { bw = w3; best = 3; }
  if (w4 < bw) { bw = w4; best = 4; }
  return best;
}

static void wgrad_tile(int cfg, int* bm, int* bn) {
  static const int t[5][2] = {{0, 0}, {128, 160}, {160, 128}, {128, 128}, {128, 32}};
  *bm = t[cfg][0]; *bn = t[cfg][1];
}

static int wgrad_plan_splitr(const ldmk_wgrad_args& a, int cfg) {
  int bm, bn;
  wgrad_tile(cfg, &bm, &bn);
  const long long tiles = (long long)((a.Kw + bm - 1) / bm) * ((a.N + bn - 1) / bn) * (a.batch > 1 ? a.batch : 1);
  const int iters = (a.R + 31) / 32;
  long long s = 512 / tiles;                       // fill, but never overflow, one round of 2 workgroups per CU ...
  if (s > iters / 8) s = iters / 8;                // ... but keep >= 8 slices per workgroup (slab write + reduce cost)
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  const long long per = (long long)(a.batch > 1 ? a.batch : 1) * (a.Kw + (a.dbias ? 1 : 0)) * a.N;
  if (s > 1 && (!a.ws || per * s > a.ws_elems)) s = a.ws ? a.ws_elems / per : 1;
  return s < 1 ? 1 : (int)s;
}

}  // namespace ldmk

extern "C" long long ldmk_wgrad_workspace_elems(const ldmk_wgrad_args* args) {
  LDMK_REQUIRE(args != nullptr, "ldmk_wgrad_workspace_elems: null args");
  ldmk_wgrad_args a = *args;
  LDMK_REQUIRE(a.R > 0 && a.Kw > 0 && a.N > 0 && a.splitr >= 0 && a.splitr <= 256, "ldmk_wgrad_workspace_elems: bad shape / splitr");
  int sr = a.splitr;
  if (sr == 0) {
    a.ws = reinterpret_cast<float*>(1);
    a.ws_elems = 1LL << 40;
    sr = ldmk::wgrad_plan_splitr(a, ldmk::wgrad_cfg(a));
  }
  if (sr <= 1) return 0;
  return (long long)(a.batch > 1 ? a.batch : 1) * sr * (a.Kw + (a.dbias ? 1 : 0)) * (long long)a.N;
}

extern "C" int ldmk_wgrad_plan(const ldmk_wgrad_args* args, int* splitr) {
  if (!args || !splitr) return LDMK_EINVAL;
  ldmk_wgrad_args a = *args;
  if (!a.ws) { a.ws = reinterpret_cast<float*>(1); a.ws_elems = 1LL << 40; }   // "how much would you like"
  *splitr = ldmk::wgrad_plan_splitr(a, ldmk::wgrad_cfg(a));
  return LDMK_OK;
}

extern "C" int ldmk_wgrad(const ldmk_wgrad_args* args, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(args != nullptr, "ldmk_wgrad: null args");
  ldmk_wgrad_args a = *args;
  LDMK_REQUIRE(a.R > 0 && a.Kw > 0 && a.N > 0 && a.a && a.dy && a.dw, "ldmk_wgrad: bad problem R=%d Kw=%d N=%d", a.R, a.Kw, a.N);
  LDMK_REQUIRE(a.N % 4 == 0 && a.ldy % 4 == 0 && a.ldw % 4 == 0 && a.Kw % 4 == 0, "ldmk_wgrad: N, Kw, ldy, ldw must be multiples of 4");
  if (a.a_mode == LDMK_A_CONV3X3) {
    LDMK_REQUIRE(a.c > 0 && a.c % 32 == 0 && a.Kw == 9 * a.c, "ldmk_wgrad: conv needs c%%32==0 and Kw == 9*c (c=%d Kw=%d)", a.c, a.Kw);
    LDMK_REQUIRE(a.in_h > 0 && a.in_w > 0 && a.out_h > 0 && a.out_w > 0 && a.stride >= 1, "ldmk_wgrad: conv geometry");
    LDMK_REQUIRE(a.R % (a.out_h * a.out_w) == 0 && a.batch <= 1, "ldmk_wgrad: R must be n*out_h*out_w, no batching");
  } else {
    LDMK_REQUIRE(a.a_mode == LDMK_A_ROWS && a.lda >= a.Kw && a.lda % 4 == 0, "ldmk_wgrad: rows mode needs lda >= Kw, lda%%4==0");
  }
  LDMK_REQUIRE(!a.dbias || a.batch <= 1, "ldmk_wgrad: dbias is not available for batched problems");
  LDMK_REQUIRE(a.splitr >= 0 && a.splitr <= 256, "ldmk_wgrad: splitr=%d outside [0,256]", a.splitr);
  LDMK_REQUIRE(a.compute == LDMK_COMPUTE_F32 || a.compute == LDMK_COMPUTE_BF16, "ldmk_wgrad: compute=%d", a.compute);
  if (a.alpha == 0.f) a.alpha = 1.f;
  const int cfg = wgrad_cfg(a);
  int sr = a.splitr > 0 ? a.splitr : wgrad_plan_splitr(a, cfg);
  if (sr > 1) {
    const long long need = (long long)(a.batch > 1 ? a.batch : 1) * sr * (a.Kw + (a.dbias ? 1 : 0)) * a.N;
    LDMK_REQUIRE_MEM(a.ws && need <= a.ws_elems, "ldmk_wgrad: splitr=%d needs a workspace of %lld floats (ldmk_wgrad_workspace_elems), "
                     "%lld given", sr, need, a.ws ? a.ws_elems : 0LL);
  }
  hipStream_t st = (hipStream_t)stream;
  switch (cfg) {
    case 1: return launch_wgrad<1, 5, 4, 1>(a, sr, st);    // 128 x 160
    case 2: return launch_wgrad<5, 1, 1, 4>(a, sr, st);    // 160 x 128
    case 3: return launch_wgrad<2, 2, 2, 2>(a, sr, st);    // 128 x 128
    default: return launch_wgrad<1, 1, 4, 1>(a, sr, st);   // 128 x 32 (per-head attention gradients)
  }
}
