// Implicit GEMM on the gfx950 f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
//   out[M][N] = epilogue( transform(A)[M][K] * W[K][N] )
//
// One kernel family serves every 3x3 / 1x1 convolution and every Linear on [n*H*W] token rows of
// the UNet and the VQGAN (see include/ldmk.h for the reference call sites).  Design:
//   * activations NHWC, so a conv tap's K-slice of 32 channels is one contiguous 128-B run per
//     pixel -> the im2col gather is 8 lanes x 16 B per row, fully coalesced;
//   * A and B K-slices are staged through LDS (A transposed to [k][m] with an odd row stride so
//     that the MFMA operand reads -- 32 consecutive m at fixed k -- and the staging writes are
//     bank-conflict-free); register prefetch of slice i+1 overlaps the MFMAs of slice i;
//   * GroupNorm(+SiLU) / LayerNorm are applied while staging A (the normalised tensor is never
//     written to HBM); channel-concat skip connections are two base pointers, never a copy;
//     nearest-x2 upsampling and stride-2 / asymmetric padding are index arithmetic in the gather;
//   * epilogue fuses bias, per-sample vector (timestep-embedding / 1-token cross-attention),
//     residual and GEGLU;
//   * tile shapes: 4 waves as WM x WN x WK, each wave TM x TN MFMA tiles of 32x32.  WK > 1 splits
//     K inside the workgroup (partials reduced through LDS in a fixed order -> deterministic),
//     which is what keeps 256 CUs busy on the 8x8-resolution layers (M = 64 rows per sample,
//     K up to 9*1280).
#include "ldmk_common.h"

namespace ldmk {

struct RowInfo {   // decomposition of one A row (output pixel) handled by this thread
  int n, oy, ox;
  bool valid;
};

template <int TM, int TN, int WM, int WN, int WK, bool BT>
__global__ __launch_bounds__(256) void igemm_kernel(const ldmk_igemm_args p) {
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  constexpr int KC = 32 * WK;           // K elements staged per iteration
  constexpr int ASTR = BM + 1;          // odd stride: conflict-free transposed writes + reads
  constexpr int BSTR = BN + (BT ? 1 : 0);
  constexpr int AROWS = BM / 32;        // float4 per thread per 32-wide K sub-chunk (A)
  constexpr int BROWS = BN / 32;        // same for B
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                     // [KC][ASTR]
  float* Bs = smem + KC * ASTR;         // [KC][BSTR]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wk = wave / (WM * WN);
  const int wm = (wave % (WM * WN)) / WN;
  const int wn = wave % WN;
  const int l31 = lane & 31, half = lane >> 5;

  const int tiles_m = (p.M + BM - 1) / BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (bid % tiles_m) * BM;
  const int n0 = (bid / tiles_m) * BN;
  const int bz = blockIdx.z;

  const float* __restrict__ a0 = p.a0 + (long long)bz * p.a_bstride;
  const float* __restrict__ a1 = p.a1;
  const float* __restrict__ wp = p.w + (long long)bz * p.w_bstride;
  float* __restrict__ outp = p.out + (long long)bz * p.out_bstride;
  const float* resp = p.residual ? p.residual + (long long)bz * p.out_bstride : nullptr;

  const int Cin = p.c0 + p.c1;
  const int cpt = Cin / 32;             // 32-channel sub-chunks per tap
  const int nkc = p.K / 32;             // total sub-chunks
  const int iters = (nkc + WK - 1) / WK;
  const bool conv = p.a_mode == LDMK_A_CONV3X3;
  const int tf = p.a_tf;

  // ---- per-thread A row bookkeeping (rows are fixed for the whole K loop)
  const int arow = tid >> 3;            // 0..31
  const int acol = (tid & 7) * 4;       // channel offset inside a 32-wide sub-chunk
  RowInfo ri[AROWS];
  float ln_mean[AROWS], ln_rstd[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    int m = m0 + arow + 32 * i;
    ri[i].valid = m < p.M;
    int mm = ri[i].valid ? m : 0;
    ri[i].n = mm / p.rows_per_sample;
    int pix = mm - ri[i].n * p.rows_per_sample;
    if (conv) {
      ri[i].oy = pix / p.out_w;
      ri[i].ox = pix - ri[i].oy * p.out_w;
    } else {
      ri[i].oy = mm;  // row index for LDMK_A_ROWS
      ri[i].ox = 0;
    }
    if (tf == LDMK_TF_LAYERNORM) {
      ln_mean[i] = p.row_stats[2 * (long long)mm];
      ln_rstd[i] = p.row_stats[2 * (long long)mm + 1];
    }
  }

  float4 areg[WK][AROWS];
  float4 breg[WK][BROWS];

  auto load_slices = [&](int it) {
#pragma unroll
    for (int j = 0; j < WK; ++j) {
      const int kc = it * WK + j;
      const bool kvalid = kc < nkc;
      // ---------------- A
      int tap = 0, cc = kc;
      if (conv) { tap = kc / cpt; cc = kc - tap * cpt; }
      const int c = cc * 32 + acol;                 // channel in the (virtual) concat
      const bool second = c >= p.c0;
      const float* src = second ? a1 : a0;
      const int cs = second ? p.c1 : p.c0;
      const int cl = second ? c - p.c0 : c;
      const int dy = tap / 3, dx = tap - dy * 3;
      float4 g4 = make_float4(0.f, 0.f, 0.f, 0.f), b4 = g4;
      if (tf == LDMK_TF_LAYERNORM && kvalid) {
        g4 = *reinterpret_cast<const float4*>(p.ln_gamma + c);
        b4 = *reinterpret_cast<const float4*>(p.ln_beta + c);
      }
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = kvalid && ri[i].valid;
        long long off = 0;
        if (conv) {
          int iy = ri[i].oy * p.stride + dy - p.pad_lo;
          int ix = ri[i].ox * p.stride + dx - p.pad_lo;
          if (p.upsample) {
            ok = ok && iy >= 0 && ix >= 0 && iy < 2 * p.in_h && ix < 2 * p.in_w;
            iy >>= 1; ix >>= 1;
          } else {
            ok = ok && iy >= 0 && ix >= 0 && iy < p.in_h && ix < p.in_w;
          }
          off = (((long long)ri[i].n * p.in_h + iy) * p.in_w + ix) * cs + cl;
        } else {
          off = (long long)ri[i].oy * cs + cl;
        }
        if (ok) {
          v = *reinterpret_cast<const float4*>(src + off);
          if (tf == LDMK_TF_AFFINE || tf == LDMK_TF_AFFINE_SILU) {
            const float* cf = p.tf_coef + ((long long)ri[i].n * 2) * Cin + c;
            float4 sc = *reinterpret_cast<const float4*>(cf);
            float4 sh = *reinterpret_cast<const float4*>(cf + Cin);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
            v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            if (tf == LDMK_TF_AFFINE_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
          } else if (tf == LDMK_TF_LAYERNORM) {
            const float mu = ln_mean[i], rs = ln_rstd[i];
            v.x = (v.x - mu) * rs * g4.x + b4.x; v.y = (v.y - mu) * rs * g4.y + b4.y;
            v.z = (v.z - mu) * rs * g4.z + b4.z; v.w = (v.w - mu) * rs * g4.w + b4.w;
          }
        }
        areg[j][i] = v;
      }
      // ---------------- B
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BT) {       // W given as [N][ldb]: rows n, contiguous k
          int n = n0 + arow + 32 * i;
          if (kvalid && n < p.N) v = *reinterpret_cast<const float4*>(wp + (long long)n * p.ldb + kc * 32 + acol);
        } else {        // W given as [K][ldb]: rows k, contiguous n
          int idx = tid + 256 * i;
          int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
          int n = n0 + n4 * 4;
          if (kvalid && n < p.N) v = *reinterpret_cast<const float4*>(wp + (long long)(kc * 32 + kk) * p.ldb + n);
        }
        breg[j][i] = v;
      }
    }
  };

  auto store_slices = [&]() {
#pragma unroll
    for (int j = 0; j < WK; ++j) {
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        float* d = As + (j * 32 + acol) * ASTR + arow + 32 * i;
        d[0] = areg[j][i].x; d[ASTR] = areg[j][i].y; d[2 * ASTR] = areg[j][i].z; d[3 * ASTR] = areg[j][i].w;
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        if (BT) {
          float* d = Bs + (j * 32 + acol) * BSTR + arow + 32 * i;
          d[0] = breg[j][i].x; d[BSTR] = breg[j][i].y; d[2 * BSTR] = breg[j][i].z; d[3 * BSTR] = breg[j][i].w;
        } else {
          int idx = tid + 256 * i;
          int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
          *reinterpret_cast<float4*>(Bs + (j * 32 + kk) * BSTR + n4 * 4) = breg[j][i];
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const float* Aw = As + (wk * 32 + half) * ASTR + wm * (32 * TM) + l31;
  const float* Bw = Bs + (wk * 32 + half) * BSTR + wn * (32 * TN) + l31;

  load_slices(0);
  for (int it = 0; it < iters; ++it) {
    __syncthreads();                 // previous iteration's MFMA reads are done
    store_slices();
    __syncthreads();
    if (it + 1 < iters) load_slices(it + 1);   // in flight while the matrix cores work
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
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }

  // ---- in-workgroup split-K reduction (fixed order -> bitwise reproducible)
  if (WK > 1) {
    __syncthreads();
    float* red = smem;  // [(WK-1)][WM*WN][TM*TN*16][64]
    constexpr int PER_WAVE = TM * TN * 16 * 64;
    if (wk > 0) {
      float* d = red + ((wk - 1) * (WM * WN) + wm * WN + wn) * PER_WAVE + lane;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) d[((i * TN + j) * 16 + r) * 64] = acc[i][j][r];
    }
    __syncthreads();
    if (wk == 0) {
#pragma unroll
      for (int q = 1; q < WK; ++q) {
        const float* s = red + ((q - 1) * (WM * WN) + wm * WN + wn) * PER_WAVE + lane;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += s[((i * TN + j) * 16 + r) * 64];
      }
    }
  }
  if (wk != 0) return;

  // ---- epilogue.  C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int rowbase = m0 + wm * (32 * TM);
  const int colbase = n0 + wn * (32 * TN);
  const float alpha = p.alpha;
  if (p.epi == LDMK_EPI_GEGLU) {
    if constexpr (TN % 2 == 0) {
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const int cv = colbase + j * 32 + l31;        // packed value column
        const int cg = cv + 32;                       // packed gate column
        if (cv >= p.N) continue;
        const int oc = ((colbase + j * 32) >> 1) + l31;
        const float bv = p.bias ? p.bias[cv] : 0.f, bg = p.bias ? p.bias[cg] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < p.M) {
              float v = acc[i][j][r] * alpha + bv;
              float g = acc[i][j + 1][r] * alpha + bg;
              float ge = 0.5f * g * (1.0f + erff(g * 0.70710678118654752440f));   // exact (erf) GELU
              outp[(long long)row * p.ldc + oc] = v * ge;
            }
          }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colbase + j * 32 + l31;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < p.M) {
          float v = acc[i][j][r] * alpha + bv;
          if (p.batch_vec) v += p.batch_vec[(long long)(row / p.rows_per_sample) * p.batch_vec_ld + col];
          long long o = (long long)row * p.ldc + col;
          if (resp) v += resp[o];
          outp[o] = v;
        }
      }
  }
}

template <int TM, int TN, int WM, int WN, int WK, bool BT>
static int launch_cfg(const ldmk_igemm_args& a, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, KC = 32 * WK;
  constexpr int ASTR = BM + 1, BSTR = BN + (BT ? 1 : 0);
  size_t stage = (size_t)KC * (ASTR + BSTR) * sizeof(float);
  size_t red = WK > 1 ? (size_t)(WK - 1) * WM * WN * TM * TN * 16 * 64 * sizeof(float) : 0;
  size_t lds = stage > red ? stage : red;
  int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  dim3 grid(tiles, 1, a.batch > 1 ? a.batch : 1);
  auto k = igemm_kernel<TM, TN, WM, WN, WK, BT>;
  static bool attr_done = false;   // per instantiation
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a);
  return check_launch("ldmk_igemm");
}

static int pick_config(const ldmk_igemm_args& a) {
  const long long b = a.batch > 1 ? a.batch : 1;
  auto nb = [&](int bm, int bn) { return b * ((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn); };
  int cfg;
  if (nb(128, 128) >= 384) cfg = 0;
  else if (a.epi == LDMK_EPI_GEGLU) cfg = nb(64, 128) >= 256 ? 1 : 2;
  else if (nb(64, 64) >= 384 || a.K < 512) cfg = 3;
  else cfg = 2;
  return cfg;
}

template <bool BT>
static int dispatch(const ldmk_igemm_args& a, hipStream_t st, int force) {
  int cfg = force >= 0 ? force : (a.tile_cfg > 0 ? a.tile_cfg - 1 : pick_config(a));
  if (a.epi == LDMK_EPI_GEGLU && cfg == 3) cfg = 2;   // GEGLU needs an even number of N tiles per wave
  switch (cfg) {
    case 0: return launch_cfg<2, 2, 2, 2, 1, BT>(a, st);   // 128x128
    case 1: return launch_cfg<1, 2, 2, 2, 1, BT>(a, st);   // 64x128
    case 2: return launch_cfg<2, 2, 1, 1, 4, BT>(a, st);   // 64x64, K split over the 4 waves
    default: return launch_cfg<1, 1, 2, 2, 1, BT>(a, st);  // 64x64
  }
}

}  // namespace ldmk

// test hook: force a tile configuration (-1 = heuristic)
static int g_force_cfg = -1;
extern "C" void ldmk_igemm_force_config(int cfg) { g_force_cfg = cfg; }

extern "C" int ldmk_igemm_pick_config(const ldmk_igemm_args* args) {
  if (!args) return LDMK_EINVAL;
  return ldmk::pick_config(*args) + 1;
}

extern "C" int ldmk_igemm(const ldmk_igemm_args* args, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(args != nullptr, "ldmk_igemm: null args");
  ldmk_igemm_args a = *args;
  LDMK_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "ldmk_igemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  LDMK_REQUIRE(a.c0 > 0 && a.c0 % 32 == 0 && a.c1 % 32 == 0, "ldmk_igemm: channel counts must be multiples of 32 (c0=%d c1=%d)", a.c0, a.c1);
  LDMK_REQUIRE((a.c1 == 0) == (a.a1 == nullptr), "ldmk_igemm: a1/c1 mismatch");
  const int taps = a.a_mode == LDMK_A_CONV3X3 ? 9 : 1;
  LDMK_REQUIRE(a.K == taps * (a.c0 + a.c1), "ldmk_igemm: K=%d != taps*(c0+c1)=%d", a.K, taps * (a.c0 + a.c1));
  LDMK_REQUIRE(a.N % 4 == 0 && a.ldb % 4 == 0, "ldmk_igemm: N and ldb must be multiples of 4");
  LDMK_REQUIRE(a.rows_per_sample > 0, "ldmk_igemm: rows_per_sample");
  if (a.a_mode == LDMK_A_CONV3X3) {
    LDMK_REQUIRE(a.in_h > 0 && a.in_w > 0 && a.out_h > 0 && a.out_w > 0 && a.stride >= 1, "ldmk_igemm: conv geometry");
    LDMK_REQUIRE(a.rows_per_sample == a.out_h * a.out_w, "ldmk_igemm: rows_per_sample != out_h*out_w");
    LDMK_REQUIRE(a.batch <= 1, "ldmk_igemm: batched conv unsupported");
  }
  if (a.a_tf == LDMK_TF_AFFINE || a.a_tf == LDMK_TF_AFFINE_SILU) LDMK_REQUIRE(a.tf_coef, "ldmk_igemm: tf_coef missing");
  if (a.a_tf == LDMK_TF_LAYERNORM) {
    LDMK_REQUIRE(a.row_stats && a.ln_gamma && a.ln_beta && a.a_mode == LDMK_A_ROWS, "ldmk_igemm: layernorm prologue args");
  }
  if (a.epi == LDMK_EPI_GEGLU) LDMK_REQUIRE(a.N % 64 == 0 && !a.residual && !a.batch_vec, "ldmk_igemm: GEGLU needs N%%64==0 and no residual");
  LDMK_REQUIRE(a.tile_cfg >= 0 && a.tile_cfg <= 4, "ldmk_igemm: tile_cfg=%d outside [0,4]", a.tile_cfg);
  if (a.alpha == 0.f) a.alpha = 1.f;
  hipStream_t st = (hipStream_t)stream;
  return a.b_trans ? dispatch<true>(a, st, g_force_cfg) : dispatch<false>(a, st, g_force_cfg);
}
