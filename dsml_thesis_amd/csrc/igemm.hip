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
//     bank-conflict-free); the raw global loads of slice i+1 are issued before the MFMAs of slice i
//     and only consumed (transformed + written to LDS) after them;
//   * LayerNorm / GroupNorm-affine are applied while staging A (the normalised tensor is never
//     written to HBM); channel-concat skip connections are two base pointers, never a copy;
//     nearest-x2 upsampling and stride-2 / asymmetric padding are index arithmetic in the gather;
//   * epilogue fuses bias, per-sample vector (timestep embedding), residual and GEGLU;
//   * tile shapes: 4 waves as WM x WN x WK, each wave TM x TN MFMA tiles of 32x32; the BN = 160
//     family matches this UNet's channel counts (multiples of model_channels = 160) with no padded
//     columns.  WK > 1 splits K inside the workgroup (LDS reduction), `splitk` > 1 splits K across
//     workgroups (partial slabs + a reduce/epilogue kernel); both sum in a fixed order, so results
//     are bitwise reproducible.  That is what keeps 256 CUs busy on the low-resolution layers
//     (M = 64 rows per sample at 8x8, K up to 9*1280).
#include "ldmk_common.h"
#include <stdlib.h>
#include <type_traits>

// Diagnostic build only (tools/igemm_probe.hip defines LDMK_IG_STAMPS): per-wave cycle totals of the main loop's phases
// (barrier 1, LDS store, barrier 2, global-load issue, MFMA block) go to args.splitk_ws (split-K off in the probe) as [wave][8] 64-bit ticks.
#if defined(LDMK_IG_STAMPS) && (LDMK_IG_STAMPS == 1 || LDMK_IG_STAMPS == 3)
#define IG_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ig_acc[i] += t_ - ig_last; ig_last = t_; } while (0)
#elif defined(LDMK_IG_STAMPS)      /* 2: only the loop's begin / end stamps (the schedule stays the shipped one) */
#define IG_T(i) do { (void)ig_acc; (void)ig_last; } while (0)
#else
#define IG_T(i) do { } while (0)
#endif

namespace ldmk {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x4 to_bf16x4(const float4& v) {          // round-to-nearest-even (v_cvt_pk_bf16_f32)
  return bf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
}

// LDMK_COMPUTE_F16X2: x' = 2^6 x = hi + lo, hi = fp16(x'), lo = fp16(x' - hi) (round-to-nearest-even; x' - hi exact in fp32)
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr float H2_A_SCALE = 64.f;            // 2^LDMK_F16X2_A_EXP
__device__ __forceinline__ void split2h(const float4& v, f16x4& h, f16x4& l) {     // v already scaled; lo: h2_lo_pair (ldmk_common.h)
  h = f16x4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
  typedef unsigned u32x2h __attribute__((ext_vector_type(2)));
  const u32x2h hu = __builtin_bit_cast(u32x2h, h);
  l = __builtin_bit_cast(f16x4, u32x2h{h2_lo_pair(hu.x, v.x, v.y), h2_lo_pair(hu.y, v.z, v.w)});
}
__device__ __forceinline__ bool h2_out_of_range(const float4& v) {                 // |x| >= LDMK_F16X2_RANGE, inf or NaN
  constexpr unsigned LIM = 0x447a0000u;       // 1000.0f
  return (__float_as_uint(v.x) & 0x7fffffffu) >= LIM || (__float_as_uint(v.y) & 0x7fffffffu) >= LIM ||
         (__float_as_uint(v.z) & 0x7fffffffu) >= LIM || (__float_as_uint(v.w) & 0x7fffffffu) >= LIM;
}

template <int I, int N, class F>
__device__ __forceinline__ void ig_static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    ig_static_for<I + 1, N>(f);
  }
}

// exact three-way split x = hi + mid + lo (each difference below is exact in fp32: the subtrahend is the leading part of x)
__device__ __forceinline__ float bf_up(__bf16 b) { return (float)b; }
__device__ __forceinline__ void split3(const float4& v, bf16x4& h, bf16x4& m, bf16x4& l) {
  h = to_bf16x4(v);
  const float4 r = make_float4(v.x - bf_up(h[0]), v.y - bf_up(h[1]), v.z - bf_up(h[2]), v.w - bf_up(h[3]));
  m = to_bf16x4(r);
  l = to_bf16x4(make_float4(r.x - bf_up(m[0]), r.y - bf_up(m[1]), r.z - bf_up(m[2]), r.w - bf_up(m[3])));
}

// BF = true: bf16 matrix-core compute for the training step (BASELINE configs[4]).  Operands stay fp32 in HBM; they are
// rounded to bf16 (RNE) while being staged, after the fp32 prologue (LayerNorm / GroupNorm-affine), into K-contiguous LDS
// rows [tile row][KC + 8] (the +8 bf16 = 16 B pad makes the 16 rows one ds_read_b128 group touches bank-disjoint), and
// multiplied with v_mfma_f32_32x32x16_bf16 (fp32 accumulate, fp32 epilogue).  Per 32-deep K slice a wave issues 2 TM TN
// bf16 MFMAs of 32 cycles instead of 16 TM TN fp32 MFMAs of 64: the matrix work shrinks 16x and the kernel becomes
// staging-bound, which is the expected regime as long as activations are stored in fp32.
//
// BF = 3: LDMK_COMPUTE_BF16X3, fp32-accurate products from six bf16 MFMAs (include/ldmk.h).  A is split three ways while it is
// staged (three bf16 images [img][tile row][KC + 8]); the weights arrive pre-split and K-contiguous (args.w_split,
// [3][N][ld] bf16), so their staging is a 16-byte copy; per 16 k a wave issues 6 TM TN MFMAs of 32 cycles where the fp32 form
// issues 8 TM TN of 64.
// ASP (X3 only): the A operand arrives PRE-SPLIT too (args.a_split: three bf16 images [3][M][ld] written once by the tensor's
// statistics pass, ldmk_ln_stats_split) -- rows mode, one source, no staging prologue.  Its staging is then a 16-byte copy like
// B's: no split arithmetic per N-tile (a GEGLU projection re-split every A element N/BN = 8..40 times), 6 + 8 loads and
// LDS stores per thread and slice instead of 4 + 8 loads, ~90 vector operations and 12 + 8 stores.
// The lean form of the epilogue below (one wave tile wholly inside M x N, no split-K, no folded LayerNorm, no
// GEGLU, 32-row tiles inside one sample): no per-element row predicate and operand branches -- in the general form every
// residual load sits in its own basic block and waits for itself -- and the operand set is a template argument: 1 = per-sample
// vector, 2 = residual, 3 = neither, 4 = both.  Same arithmetic, rounding by rounding (csrc/igemm_ps.hip has the same pair of forms).
__device__ __forceinline__ float ig_col_finish(float acc_alpha, float bias) {
#pragma clang fp contract(off)
  return acc_alpha + bias;
}
__device__ __forceinline__ float ig_col_finish(float acc_alpha, float bias, float extra) {
#pragma clang fp contract(off)
  const float t = acc_alpha + bias;
  return t + extra;
}
__device__ __forceinline__ float ig_col_finish(float acc_alpha, float bias, float vec, float res) {
#pragma clang fp contract(off)
  float t = acc_alpha + bias;
  t = t + vec;
  return t + res;
}
template <int TM, int TN, int LEAN>
__device__ __forceinline__ void ig_lean_epilogue(const ldmk_igemm_args& p, f32x16 (&acc)[TM][TN], const int rowbase, const int colbase, const int bz,
                                                 const int l31, const int half) {
  float* __restrict__ outp = p.out + (long long)bz * p.out_bstride;
  const float* __restrict__ resp = (LEAN == 2 || LEAN == 4) ? p.residual + (long long)bz * p.out_bstride : nullptr;
  const float alpha = p.alpha;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colbase + j * 32 + l31;
    const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int r0 = rowbase + i * 32 + 4 * half;
      const unsigned obase = (unsigned)r0 * (unsigned)p.ldc + (unsigned)col;
      float extra[16], vec = 0.f;
      if constexpr (LEAN == 1 || LEAN == 4) vec = p.batch_vec[(long long)((rowbase + i * 32) / p.rows_per_sample) * p.batch_vec_ld + col];
      if constexpr (LEAN == 2 || LEAN == 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) extra[r] = resp[obase + (unsigned)(((r & 3) + 8 * (r >> 2)) * p.ldc)];
      }
      float vals[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = acc[i][j][r] * alpha;
        if constexpr (LEAN == 3) vals[r] = ig_col_finish(t, bv);
        else if constexpr (LEAN == 1) vals[r] = ig_col_finish(t, bv, vec);
        else if constexpr (LEAN == 2) vals[r] = ig_col_finish(t, bv, extra[r]);
        else vals[r] = ig_col_finish(t, bv, vec, extra[r]);
        outp[obase + (unsigned)(((r & 3) + 8 * (r >> 2)) * p.ldc)] = vals[r];
      }
      if (p.stats_out) {
        const float shift = __shfl(vals[0], l31, 64);      // row 0 of the tile
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = vals[r] - shift;
          sm += d;
          sq = fmaf(d, d, sq);
        }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (half == 0) {
          float* d = p.stats_out + ((long long)((rowbase + i * 32) >> 5) * p.N + col) * 3;
          d[0] = shift; d[1] = sm; d[2] = sq;
        }
      }
    }
  }
}

template <int TM, int TN, int WM, int WN, int WK, int KS, bool DB, bool BT, int BF = 0, bool FG = true, bool ASP = false>
__global__ __launch_bounds__(256) void igemm_kernel(const ldmk_igemm_args p, const int splitk, float* __restrict__ ws, const int nfast) {
  constexpr int BM = 32 * TM * WM;
  constexpr int BN = 32 * TN * WN;
  // BF = 2: plain bf16 compute (one image) with the weights PRE-PACKED like the split form's -- bf16, transposed, K-contiguous
  // (ldmk_pack_wbf16t, once per optimiser step): the training step's forward GEMMs.  BF = 1 reads fp32 W as it lies and gathers
  // every B fragment with eight ds_read_b32 + conversions; here a fragment is one ds_read_b128.
  constexpr bool X3 = BF >= 2;               // (named for the split form; "pre-packed bf16 weight images" is what it selects)
  // BF = 4: LDMK_COMPUTE_F16X2 -- the structure of BF = 3 with TWO fp16 images per operand (A scaled by 2^6 and split while
  // staged, range-checked; W pre-split by ldmk_pack_wsplit_h2) and THREE matrix instructions per product.
  constexpr bool H2 = BF == 4;
  constexpr int NI = BF == 3 ? 3 : (H2 ? 2 : 1);        // 16-bit images per operand
  static_assert(!ASP || BF == 3, "a pre-split A operand belongs to the bf16x3 form");
  constexpr int ASI = ASP ? (3 * BM * 4 + 255) / 256 : 1;    // ASP: 16-byte items of the three A images per thread and 32-k slice
  constexpr int BSI = X3 ? (NI * BN * 4 + 255) / 256 : 1;     // X3: 16-byte items of the three B images per thread and 32-k slice
  constexpr int NS = WK * KS;           // 32-wide K slices staged per iteration (KS per wave-group)
  constexpr int KC = 32 * NS;           // K elements staged per iteration
  constexpr int ASTR = BM + 1;          // odd stride: conflict-free transposed writes + reads
  constexpr int BSTR = BN + (BT ? 1 : 0);
  constexpr int AROWS = BM / 32;        // float4 per thread per 32-wide K sub-chunk (A)
  constexpr int BROWS = BN / 32;        // same for B
  constexpr int RS = KC + 8;            // BF: bf16 elements per LDS row
  static_assert(WM * WN * WK == 4, "4 waves per workgroup");
  static_assert(!BF || (WK == 1 && !DB), "the bf16 forms are built for WK = 1, single-buffered");
  static_assert(!X3 || (!BT && FG), "the split form reads pre-split W (b_trans = 0) through the fast gather");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int STAGE = KC * (ASTR + BSTR);   // floats per staging buffer (DB: two of them)
  float* As = smem;                     // [KC][ASTR]
  float* Bs = smem + KC * ASTR;         // [KC][BSTR]
  __bf16* As16 = reinterpret_cast<__bf16*>(smem);               // BF: [BM][RS] bf16, K-contiguous rows
  __bf16* Bs16 = reinterpret_cast<__bf16*>(smem) + BM * RS;     // BF, BT: [BN][RS] bf16 (W^T rows are K-contiguous in HBM)
  float* Bs32 = smem + BM * RS / 2;                             // BF, !BT: [KC][BN] fp32 as it lies in HBM (n-contiguous);
                                                                // fragments are gathered from it, see compute()
  constexpr int AIMG = BM * RS, BIMG = BN * RS;                 // X3: elements per image; A images first, then the B images
  __bf16* Bx16 = reinterpret_cast<__bf16*>(smem) + NI * AIMG;

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
  // (nfast, round 5: the column tiles of one row tile adjacent on an XCD -- they share the A tile in its L2; csrc/igemm_ps.hip)
  const int m0 = (nfast & 1) ? (bid / tiles_n) * BM : (bid % tiles_m) * BM;
  const int n0 = (nfast & 1) ? (bid % tiles_n) * BN : (bid / tiles_m) * BN;
  const int ks = blockIdx.y;            // cross-workgroup K split index
  const int bz = blockIdx.z;

  const float* __restrict__ a0 = p.a0 + (long long)bz * p.a_bstride;
  const float* __restrict__ a1 = p.a1;
  const float* __restrict__ wp = p.w + (long long)bz * p.w_bstride;

  const int Cin = p.c0 + p.c1;
  const int nkc = p.K / 32;             // total sub-chunks
  const int iters_all = (nkc + NS - 1) / NS;
  const int it_per = (iters_all + splitk - 1) / splitk;
  const int it_begin = ks * it_per;
  const int it_end = min(iters_all, it_begin + it_per);
  const bool conv = p.a_mode == LDMK_A_CONV3X3;
  const int tf = p.a_tf == LDMK_TF_LAYERNORM_FOLDED ? LDMK_TF_NONE : p.a_tf;   // folded LayerNorm: raw rows here, the
                                                                               // two per-row scalars in the epilogue

  // ---- per-thread A row bookkeeping (rows are fixed for the whole K loop); 32-bit element indices
  const int arow = tid >> 3;            // 0..31
  const int acol = (tid & 7) * 4;       // channel offset inside a 32-wide sub-chunk
  int r_n[AROWS], r_y[AROWS], r_x[AROWS];
  unsigned r_mask[AROWS];               // conv: bit t set <=> tap t reads inside the image
  float ln_mean[AROWS], ln_rstd[AROWS];
#pragma unroll
  for (int i = 0; i < AROWS; ++i) {
    const int m = m0 + arow + 32 * i;
    const bool valid = m < p.M;
    const int mm = valid ? m : 0;
    r_n[i] = mm / p.rows_per_sample;
    if (conv) {
      const int pix = mm - r_n[i] * p.rows_per_sample;
      const int oy = pix / p.out_w, ox = pix - oy * p.out_w;
      r_y[i] = oy * p.stride - p.pad_lo;
      r_x[i] = ox * p.stride - p.pad_lo;
      unsigned mask = 0;
      const int lim_h = p.upsample ? 2 * p.in_h : p.in_h, lim_w = p.upsample ? 2 * p.in_w : p.in_w;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = r_y[i] + t / 3, ix = r_x[i] + t % 3;
        // upsample == 2: zero-insertion (data gradient of a stride-2 convolution): odd positions are zeros
        if (valid && iy >= 0 && ix >= 0 && iy < lim_h && ix < lim_w && !(p.upsample == 2 && ((iy | ix) & 1))) mask |= 1u << t;
      }
      r_mask[i] = mask;
    } else {
      r_y[i] = mm;
      r_x[i] = 0;
      r_mask[i] = valid ? 1u : 0u;
    }
    if (tf == LDMK_TF_LAYERNORM) {
      ln_mean[i] = p.row_stats[2 * (long long)mm];
      ln_rstd[i] = p.row_stats[2 * (long long)mm + 1];
    }
  }

  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  float4 areg[NS][AROWS];
  float4 breg[X3 ? 1 : NS][X3 ? 1 : BROWS];
  u32x4_t bxreg[NS][BSI];               // X3: the next slice's pre-split weight items
  u32x4_t axreg[NS][ASI];               // ASP: ... and pre-split activation items

  // ---- fast gather (every launch except the upsampling convolutions).  Measured with s_memtime stamps
  // (tools/igemm_probe.hip) on the ResBlock convolutions: of 7400 cycles per 32-deep slice the generic gather below spent
  // 1225 issuing its 9 loads (~300 VALU instructions of index arithmetic, 64-bit addresses and one divergent branch per
  // masked load) -- 3000 when the co-resident wave was in its MFMA block, because non-MFMA instructions compete for the
  // SIMD's issue slots.  Here everything that does not change along K is computed once: per row a 32-bit byte offset
  // of (tap 0, channel acol) in each source, per B item its offset in the first slice; per slice only scalars change
  // (tap, channel chunk -> one SGPR byte offset), so a load costs one add and the tap-mask select.  Loads are raw buffer
  // loads: a masked tap / ragged column gets offset 0xFFFFFFFF, which is out of range and returns zeros -- no branch.
  const long long samples_ = ((long long)p.M + p.rows_per_sample - 1) / p.rows_per_sample;
  const long long a_rows_ = conv ? samples_ * p.in_h * p.in_w : (long long)p.M;
  const long long w_bytes_ = (BT ? (long long)p.N : (long long)p.K) * p.ldb * 4;
  constexpr bool fastg = FG;       // the host picks the instantiation (igemm_fast_gather_ok): no zero-insertion, sources < 4 GB
  unsigned aoff0[AROWS], aoff1[AROWS], boff[BROWS];
  __amdgpu_buffer_rsrc_t rs_a0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a0), 0, (int)(unsigned)(a_rows_ * p.c0 * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a1 ? a1 : a0), 0, (int)(unsigned)(a_rows_ * p.c1 * 4), 0x00020000);
  __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wp), 0, (int)(unsigned)w_bytes_, 0x00020000);
  // nearest-x2 upsampling folded into the gather (upsample == 1, single source): tap (dy, dx) of output pixel (oy, ox)
  // reads input pixel ((oy - 1 + dy) >> 1, (ox - 1 + dx) >> 1).  Relative to the pixel of tap (0, 0) that is a row step of
  // {0, e, 1} for dy = {0, 1, 2} with e = 1 for even oy and 0 for odd oy (same in x): the parity-dependent middle steps
  // are two more per-row byte offsets (aoff1 doubles as the x one: the second source does not exist in this mode).
  unsigned upy[AROWS];
  unsigned bxoff[BSI], bxlds[BSI];
  __amdgpu_buffer_rsrc_t rs_wx = rs_w;
  if constexpr (X3) {
    const __bf16* wx = reinterpret_cast<const __bf16*>(p.w_split) + (long long)bz * p.w_split_bstride;
    rs_wx = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wx), 0, (int)(unsigned)((long long)NI * p.N * p.w_split_ld * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < BSI; ++i) {
      const int idx = tid + 256 * i;
      const int img = idx / (BN * 4), rem = idx - img * (BN * 4);
      const int nn = rem >> 2, q = rem & 3;
      const bool ok = idx < NI * BN * 4 && n0 + nn < p.N;
      bxoff[i] = ok ? (unsigned)((((long long)img * p.N + n0 + nn) * p.w_split_ld + q * 8) * 2) : 0xFFFFFFFFu;
      bxlds[i] = (unsigned)((img * BIMG + nn * RS + q * 8) * 2);                  // byte offset inside the B images
    }
  }
  unsigned axoff[ASI], axlds[ASI];
  __amdgpu_buffer_rsrc_t rs_ax = rs_w;
  if constexpr (ASP) {
    const __bf16* ax = reinterpret_cast<const __bf16*>(p.a_split);
    rs_ax = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(ax), 0, (int)(unsigned)(3LL * p.M * p.a_split_ld * 2), 0x00020000);
#pragma unroll
    for (int i = 0; i < ASI; ++i) {
      const int idx = tid + 256 * i;
      const int img = idx / (BM * 4), rem = idx - img * (BM * 4);
      const int rr = rem >> 2, q = rem & 3;
      const bool ok = idx < 3 * BM * 4 && m0 + rr < p.M;
      axoff[i] = ok ? (unsigned)((((long long)img * p.M + m0 + rr) * p.a_split_ld + q * 8) * 2) : 0xFFFFFFFFu;
      axlds[i] = (unsigned)((img * AIMG + rr * RS + q * 8) * 2);                   // byte offset inside the A images
    }
  }
  if constexpr (FG) {
#pragma unroll
    for (int i = 0; i < AROWS; ++i) {
      // conv: pixel of tap (0,0) (may lie outside the image for border rows: only masked taps use it); rows: the row
      long long pix = conv ? ((long long)r_n[i] * p.in_h + r_y[i]) * p.in_w + r_x[i] : (long long)r_y[i];
      if (p.upsample == 1) {      // r_y / r_x are coordinates in the upsampled grid (oy - 1, ox - 1): floor-halve them
        const int by = r_y[i] >> 1, bx = r_x[i] >> 1;       // arithmetic shift: -1 -> -1 (masked taps only)
        pix = ((long long)r_n[i] * p.in_h + by) * p.in_w + bx;
        upy[i] = (r_y[i] & 1) ? (unsigned)(p.in_w * p.c0 * 4) : 0u;     // oy even <=> r_y odd: the middle tap steps one row
        aoff1[i] = (r_x[i] & 1) ? (unsigned)(p.c0 * 4) : 0u;
        aoff0[i] = (unsigned)((pix * p.c0 + acol) * 4);
        continue;
      }
      upy[i] = 0u;
      aoff0[i] = (unsigned)((pix * p.c0 + acol) * 4);       // mod 2^32: offset + per-slice scalar is exact for valid taps
      aoff1[i] = (unsigned)((pix * p.c1 + acol) * 4);
    }
#pragma unroll
    for (int i = 0; i < BROWS; ++i) {
      if (BT) {
        const int n = n0 + arow + 32 * i;
        boff[i] = n < p.N ? (unsigned)(((long long)n * p.ldb + acol) * 4) : 0xFFFFFFFFu;
      } else {
        const int idx = tid + 256 * i;
        const int kk = idx / (BN / 4), n = n0 + (idx - kk * (BN / 4)) * 4;
        boff[i] = n < p.N ? (unsigned)(((long long)kk * p.ldb + n) * 4) : 0xFFFFFFFFu;
      }
    }
  }
  auto bload = [&](__amdgpu_buffer_rsrc_t rs, unsigned off) -> float4 {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
  };
  auto load_slices_fast = [&](int it) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int kc = it * NS + j;
      const bool kvalid = kc < nkc;
      int tap = 0, cc = kc;
      if (conv) { cc = kc / 9; tap = kc - cc * 9; }
      const bool second = cc * 32 >= p.c0;                   // c0 % 32 == 0: a 32-channel chunk lies in one source
      const int cs = second ? p.c1 : p.c0;
      const int dy = tap / 3, dx = tap - dy * 3;
      const bool ups = p.upsample == 1;
      // upsampling: taps 0 / 2 step 0 / 1 input pixels, tap 1 steps by the row's parity (upy / aoff1, added below)
      const int sy = ups ? (dy == 2) : dy, sx = ups ? (dx == 2) : dx;
      const unsigned sa = (unsigned)(((conv ? (sy * p.in_w + sx) * cs : 0) + cc * 32 - (second ? p.c0 : 0)) * 4);
      const unsigned sb = (unsigned)((BT ? (long long)kc * 32 : (long long)kc * 32 * p.ldb) * 4);
      const unsigned tbit = 1u << tap;
      // all offsets first, then the loads back to back: when the compiler interleaves them it re-uses the destination
      // registers of the previous slice's loads as temporaries and puts a conservative s_waitcnt vmcnt(1) -- a full
      // memory round trip -- in the middle of the issue sequence
      unsigned oa[AROWS], ob[BROWS];
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        unsigned o_ = (second ? aoff1[i] : aoff0[i]) + sa;
        if (ups) o_ = aoff0[i] + sa + (dy == 1 ? upy[i] : 0u) + (dx == 1 ? aoff1[i] : 0u);
        oa[i] = (kvalid && (r_mask[i] & tbit)) ? o_ : 0xFFFFFFFFu;
      }
      if constexpr (X3) {
        unsigned ox[BSI];
#pragma unroll
        for (int i = 0; i < BSI; ++i) ox[i] = (kvalid && bxoff[i] != 0xFFFFFFFFu) ? bxoff[i] + (unsigned)(kc * 64) : 0xFFFFFFFFu;
        if constexpr (ASP) {
          unsigned oy_[ASI];
#pragma unroll
          for (int i = 0; i < ASI; ++i) oy_[i] = (kvalid && axoff[i] != 0xFFFFFFFFu) ? axoff[i] + (unsigned)(kc * 64) : 0xFFFFFFFFu;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < BSI; ++i) bxreg[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wx, (int)ox[i], 0, 0);
#pragma unroll
          for (int i = 0; i < ASI; ++i) axreg[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_ax, (int)oy_[i], 0, 0);
        } else {
          // the weight items FIRST: loads return in order, so the store phase can copy B to LDS (vmcnt = the A loads still in
          // flight) while the A rows -- the gather, the longer latency -- are still arriving
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < BSI; ++i) bxreg[j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs_wx, (int)ox[i], 0, 0);
#pragma unroll
          for (int i = 0; i < AROWS; ++i) areg[j][i] = second ? bload(rs_a1, oa[i]) : bload(rs_a0, oa[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < BROWS; ++i) ob[i] = (kvalid && boff[i] != 0xFFFFFFFFu) ? boff[i] + sb : 0xFFFFFFFFu;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < AROWS; ++i) areg[j][i] = second ? bload(rs_a1, oa[i]) : bload(rs_a0, oa[i]);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) breg[j][i] = bload(rs_w, ob[i]);
      }
    }
  };

  // generic gather: nearest-x2 upsampling folded into the index math (Upsample convolutions), zero-inserted x2 (data
  // gradient of a stride-2 convolution), tensors beyond 4 GB
  auto load_slices_generic = [&](int it) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const int kc = it * NS + j;
      const bool kvalid = kc < nkc;
      int tap = 0, cc = kc;
      if (conv) { cc = kc / 9; tap = kc - cc * 9; }  // chunk-major, tap-minor: 9 consecutive slices re-read the
      const int c = cc * 32 + acol;                 // same pixels' 128-B runs (L1/L2 reuse); channel in the concat
      const bool second = c >= p.c0;
      const float* src = second ? a1 : a0;
      const int cs = second ? p.c1 : p.c0;
      const int cl = second ? c - p.c0 : c;
      const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kvalid && ((r_mask[i] >> tap) & 1u)) {
          unsigned off;
          if (conv) {
            int iy = r_y[i] + dy, ix = r_x[i] + dx;
            if (p.upsample) { iy >>= 1; ix >>= 1; }
            off = ((unsigned)(r_n[i] * p.in_h + iy) * p.in_w + ix) * cs + cl;
          } else {
            off = (unsigned)r_y[i] * cs + cl;
          }
          v = *reinterpret_cast<const float4*>(src + off);
        }
        areg[j][i] = v;
      }
#pragma unroll
      for (int i = 0; i < (X3 ? 0 : BROWS); ++i) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BT) {       // W given as [N][ldb]: rows n, contiguous k
          const int n = n0 + arow + 32 * i;
          if (kvalid && n < p.N) v = *reinterpret_cast<const float4*>(wp + (long long)n * p.ldb + kc * 32 + acol);
        } else {        // W given as [K][ldb]: rows k, contiguous n
          const int idx = tid + 256 * i;
          const int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
          const int n = n0 + n4 * 4;
          if (kvalid && n < p.N) v = *reinterpret_cast<const float4*>(wp + (long long)(kc * 32 + kk) * p.ldb + n);
        }
        breg[j][i] = v;
      }
    }
  };

#ifdef LDMK_IG_STAMPS
  unsigned long long ig_acc[6] = {0, 0, 0, 0, 0, 0}, ig_last = 0;
#endif
  bool h2_bad = false;          // H2: a staged A element outside the scaled fp16 range
  // A-side prologue (norm) on the loaded registers, then registers -> LDS
  auto store_slices = [&](int it, int boff) {
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      if (tf != LDMK_TF_NONE) {
        const int kc = it * NS + j;
        int tap = 0, cc = kc;
        if (conv) { cc = kc / 9; tap = kc - cc * 9; }
        const int c = cc * 32 + acol;
        if (kc < nkc) {
          if (tf == LDMK_TF_LAYERNORM) {
            const float4 g4 = *reinterpret_cast<const float4*>(p.ln_gamma + c);
            const float4 b4 = *reinterpret_cast<const float4*>(p.ln_beta + c);
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
              float4 v = areg[j][i];
              const float mu = ln_mean[i], rs = ln_rstd[i];
              v.x = (v.x - mu) * rs * g4.x + b4.x; v.y = (v.y - mu) * rs * g4.y + b4.y;
              v.z = (v.z - mu) * rs * g4.z + b4.z; v.w = (v.w - mu) * rs * g4.w + b4.w;
              areg[j][i] = (r_mask[i] & 1u) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
          } else {
#pragma unroll
            for (int i = 0; i < AROWS; ++i) {
              if ((r_mask[i] >> tap) & 1u) {     // padded taps stay exactly zero
                const float* cf = p.tf_coef + ((long long)r_n[i] * 2) * Cin + c;
                const float4 sc = *reinterpret_cast<const float4*>(cf);
                const float4 sh = *reinterpret_cast<const float4*>(cf + Cin);
                float4 v = areg[j][i];
                v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
                v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                if (tf == LDMK_TF_AFFINE_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                areg[j][i] = v;
              }
            }
          }
        }
      }
      if constexpr (ASP) {
#pragma unroll
        for (int i = 0; i < ASI; ++i) {
          if (tid + 256 * i < 3 * BM * 4)
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(As16) + axlds[i] + j * 64) = axreg[j][i];
        }
#pragma unroll
        for (int i = 0; i < BSI; ++i) {
          if (tid + 256 * i < NI * BN * 4)
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(Bx16) + bxlds[i] + j * 64) = bxreg[j][i];
        }
        continue;
      }
      if constexpr (X3) {
#if defined(LDMK_IG_STAMPS) && LDMK_IG_STAMPS == 3      /* probe: the wait for the slice's global loads goes to phase 0 */
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        IG_T(0);
#endif
#pragma unroll
        for (int i = 0; i < BSI; ++i) {                 // B first: its loads were issued first (see load_slices_fast)
          if (tid + 256 * i < NI * BN * 4)
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(Bx16) + bxlds[i] + j * 64) = bxreg[j][i];
        }
#pragma unroll
        for (int i = 0; i < AROWS; ++i) {
          __bf16* d = As16 + (arow + 32 * i) * RS + j * 32 + acol;
          if constexpr (NI == 3) {
            bf16x4 h, m, l;
            split3(areg[j][i], h, m, l);
            *reinterpret_cast<bf16x4*>(d) = h;
            *reinterpret_cast<bf16x4*>(d + AIMG) = m;
            *reinterpret_cast<bf16x4*>(d + 2 * AIMG) = l;
          } else if constexpr (H2) {
            float4 v = areg[j][i];
            if (__builtin_expect(h2_out_of_range(v), 0)) {      // a real branch (never taken on a healthy model): saturate, see h2_clamp
              asm volatile("; f16x2 operand out of range");
              h2_bad = true;
              v = h2_clamp4(v);
            }
            f16x4 h, l;
            split2h(make_float4(v.x * H2_A_SCALE, v.y * H2_A_SCALE, v.z * H2_A_SCALE, v.w * H2_A_SCALE), h, l);
            *reinterpret_cast<f16x4*>(d) = h;
            *reinterpret_cast<f16x4*>(d + AIMG) = l;
          } else {
            *reinterpret_cast<bf16x4*>(d) = to_bf16x4(areg[j][i]);
          }
        }
        continue;
      }
      if (BF) {
#pragma unroll
        for (int i = 0; i < AROWS; ++i)
          *reinterpret_cast<bf16x4*>(As16 + (arow + 32 * i) * RS + j * 32 + acol) = to_bf16x4(areg[j][i]);
#pragma unroll
        for (int i = 0; i < BROWS; ++i) {
          if (BT) {
            *reinterpret_cast<bf16x4*>(Bs16 + (arow + 32 * i) * RS + j * 32 + acol) = to_bf16x4(breg[j][i]);
          } else {
            const int idx = tid + 256 * i;
            const int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
            *reinterpret_cast<float4*>(Bs32 + (j * 32 + kk) * BN + n4 * 4) = breg[j][i];
          }
        }
        continue;
      }
#pragma unroll
      for (int i = 0; i < AROWS; ++i) {
        float* d = As + boff + (j * 32 + acol) * ASTR + arow + 32 * i;
        d[0] = areg[j][i].x; d[ASTR] = areg[j][i].y; d[2 * ASTR] = areg[j][i].z; d[3 * ASTR] = areg[j][i].w;
      }
#pragma unroll
      for (int i = 0; i < BROWS; ++i) {
        if (BT) {
          float* d = Bs + boff + (j * 32 + acol) * BSTR + arow + 32 * i;
          d[0] = breg[j][i].x; d[BSTR] = breg[j][i].y; d[2 * BSTR] = breg[j][i].z; d[3 * BSTR] = breg[j][i].w;
        } else {
          const int idx = tid + 256 * i;
          const int kk = idx / (BN / 4), n4 = idx - kk * (BN / 4);
          *reinterpret_cast<float4*>(Bs + boff + (j * 32 + kk) * BSTR + n4 * 4) = breg[j][i];
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

  const float* Aw = As + (wk * KS * 32 + half) * ASTR + wm * (32 * TM) + l31;
  const float* Bw = Bs + (wk * KS * 32 + half) * BSTR + wn * (32 * TN) + l31;

  auto compute = [&](int boff) {
    if constexpr (X3) {
      const __bf16* Aw16 = As16 + (wm * (32 * TM) + l31) * RS + 8 * half;
      const __bf16* Bw16 = Bx16 + (wn * (32 * TN) + l31) * RS + 8 * half;
      // Operand fragments are double-buffered in registers and the reads of step t + 1 are pinned in front of the matrix
      // instructions of step t (sched_group_barrier): the first version read b8, waited (lgkmcnt(0)) and multiplied, a dozen
      // exposed LDS round trips per 16 k -- 2734 cycles in the MFMA block for 1920 of matrix work (s_memtime stamps).
      constexpr int NSTEP = (KC / 16) * TN;          // one step = one B column tile of one 16-deep k group: 6 TM (1 TM) matrix instructions
      constexpr int NMM = NI == 3 ? 6 : (NI == 2 ? 3 : 1);
      bf16x8 a8[2][NI][TM], b8[2][NI];
      auto fa = [&](int s, int q) {
#pragma unroll
        for (int g = 0; g < NI; ++g)
#pragma unroll
          for (int i = 0; i < TM; ++i) a8[q][g][i] = *reinterpret_cast<const bf16x8*>(Aw16 + g * AIMG + i * 32 * RS + 16 * s);
      };
      auto fb = [&](int t, int q) {
        const int s = t / TN, j = t - s * TN;
#pragma unroll
        for (int g = 0; g < NI; ++g) b8[q][g] = *reinterpret_cast<const bf16x8*>(Bw16 + g * BIMG + j * 32 * RS + 16 * s);
      };
      fa(0, 0);
      fb(0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, NI * TM + NI, 0);       // (the first step's own reads: they open the sequence)
      ig_static_for<0, NSTEP>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int s = t / TN, j = t - s * TN;
        constexpr bool more = t + 1 < NSTEP, newk = more && (t + 1) % TN == 0;
        constexpr int nread = more ? NI + (newk ? NI * TM : 0) : 0;
        if constexpr (more) fb(t + 1, (t + 1) & 1);
        if constexpr (newk) fa((t + 1) / TN, ((t + 1) / TN) & 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (H2) {
            // lo hi, hi lo, hi hi (images: 0 = hi, 1 = lo)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a8[s & 1][1][i]), __builtin_bit_cast(f16x8, b8[t & 1][0]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a8[s & 1][0][i]), __builtin_bit_cast(f16x8, b8[t & 1][1]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a8[s & 1][0][i]), __builtin_bit_cast(f16x8, b8[t & 1][0]), acc[i][j], 0, 0, 0);
            continue;
          }
          if constexpr (NI == 3) {
            // smallest partial products first (images: 0 = hi, 1 = mid, 2 = lo)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s & 1][2][i], b8[t & 1][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s & 1][0][i], b8[t & 1][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s & 1][1][i], b8[t & 1][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s & 1][1][i], b8[t & 1][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s & 1][0][i], b8[t & 1][1], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[s & 1][0][i], b8[t & 1][0], acc[i][j], 0, 0, 0);
        }
        if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);     // the next step's LDS reads ...
        __builtin_amdgcn_sched_group_barrier(0x008, NMM * TM, 0);                             // ... then this step's matrix instructions
        // (Also measured, A/B on one box: the loads issued BEFORE the second barrier, as soon as the registers are free --
        //  -2.9 % with __syncthreads (a fence: it drains them) and -2.5 % with LDS-only barriers (s_waitcnt lgkmcnt(0); s_barrier);
        //  LDS-only barriers alone +0.15 %, not kept.)
        // (The next slice's global loads stay in ONE burst in front of this block.  Spreading them behind the steps with
        //  sched_group_barrier(0x020, ...), which pays in the f32 form, costs 2.2 % here -- 686 -> 671 sample-steps/s, A/B on one
        //  box: this form waits ~1200 cycles per slice for those loads as it is, and later issue means later arrival.)
      });
      return;
    }
    if constexpr (BF) {
      const __bf16* Aw16 = As16 + (wm * (32 * TM) + l31) * RS + 8 * half;
      const __bf16* Bw16 = Bs16 + (wn * (32 * TN) + l31) * RS + 8 * half;
#pragma unroll
      for (int s = 0; s < KC / 16; ++s) {
        bf16x8 a8[TM], b8[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a8[i] = *reinterpret_cast<const bf16x8*>(Aw16 + i * 32 * RS + 16 * s);
        if (BT) {
#pragma unroll
          for (int j = 0; j < TN; ++j) b8[j] = *reinterpret_cast<const bf16x8*>(Bw16 + j * 32 * RS + 16 * s);
        } else {
          // W is [K][N] in HBM: a K-contiguous bf16 image would need 2-byte scatter stores or 4 dword loads per thread
          // (tried: the staging instructions cost more than the matrix work they feed).  Instead the fp32 slice is staged
          // as it lies and the 8 k of a fragment are gathered with strided ds_read_b32 (consecutive lanes = consecutive
          // columns: conflict-free) and rounded in registers.
          const float* Bg = Bs32 + (16 * s + 8 * half) * BN + wn * (32 * TN) + l31;
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) b8[j][e] = (__bf16)Bg[e * BN + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
      }
      return;
    }
    // operands of k-step s+1 are read from LDS before the MFMAs of step s are issued (two register sets)
    float a[2][TM], b[2][TN];
    auto fetch = [&](int s, int q) {
#pragma unroll
      for (int i = 0; i < TM; ++i) a[q][i] = Aw[boff + 2 * s * ASTR + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[q][j] = Bw[boff + 2 * s * BSTR + j * 32];
    };
    fetch(0, 0);
#pragma unroll
    for (int s = 0; s < 16 * KS; ++s) {
      if (s + 1 < 16 * KS) fetch(s + 1, (s + 1) & 1);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][i], b[s & 1][j], acc[i][j], 0, 0, 0);
      // pin the interleave: the LDS reads of the next step go out ahead of this step's MFMAs, so their latency is
      // covered by TM*TN matrix instructions instead of being waited for in front of every pair of them
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
      // ... and ONE of the next slice's global loads behind each of the first k-steps: issued together in front of the
      // MFMA block, the 4 waves of a workgroup queue 36 KiB on the CU's 64 B/clk vector-memory path at the same moment
      // (~800 cycles per slice per wave in which no MFMA issues, measured with tools/igemm_probe.hip)
      if (s < NS * (AROWS + BROWS)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
    }
  };

  auto load_slices = [&](int it) {
    if constexpr (FG) load_slices_fast(it);
    else load_slices_generic(it);
  };
  if (it_begin < it_end) load_slices(it_begin);
  if (DB) {
    // two LDS buffers: slice i+1 is written while nobody reads its buffer -> one barrier per slice.
    // Measured on MI355X (round 1): 5 % SLOWER than the two-barrier single buffer on every tile shape (the
    // doubled LDS footprint and the later store placement cost more than the barrier saves at 2 waves/SIMD),
    // so no shipped configuration enables it; kept for re-evaluation with deeper pipelines.
    int cur = 0;
    if (it_begin < it_end) {
      store_slices(it_begin, 0);
      __syncthreads();
    }
    for (int it = it_begin; it < it_end; ++it) {
      const bool more = it + 1 < it_end;
      if (more) load_slices(it + 1);              // global loads in flight under the MFMAs
      compute(cur * STAGE);
      if (more) store_slices(it + 1, (cur ^ 1) * STAGE);
      __syncthreads();
      cur ^= 1;
    }
  } else {
    // one copy of the loop per gather form, so that the loads and the MFMA block are ONE basic block (scheduling region)
    auto run_loop = [&](auto&& loader) {
#ifdef LDMK_IG_STAMPS
      ig_last = __builtin_amdgcn_s_memtime();
      const unsigned long long ig_t0 = ig_last, ig_rt0 = __builtin_amdgcn_s_memrealtime();    // 100 MHz reference
#endif
      for (int it = it_begin; it < it_end; ++it) {
        // (s_setprio(2) around the staging phase, so that its instructions win issue slots over the co-resident wave's
        // MFMA stream: measured neutral to -2 %, not kept)
        __syncthreads();                 // previous iteration's MFMA reads are done
        IG_T(0);
        store_slices(it, 0);
        IG_T(1);
        __syncthreads();
        IG_T(2);
        loader(min(it + 1, it_end - 1));   // in flight while the matrix cores work; unconditional (the last slice re-loads
        IG_T(3);                           // itself) so that the loads share the MFMA block's scheduling region
        compute(0);
        IG_T(4);
      }
#ifdef LDMK_IG_STAMPS
      if (lane == 0) {
        unsigned long long* d = reinterpret_cast<unsigned long long*>(ws) + ((long long)blockIdx.x * 4 + wave) * 8;
        for (int q = 0; q < 5; ++q) d[q] = ig_acc[q];
        d[5] = ig_t0; d[6] = __builtin_amdgcn_s_memtime(); d[7] = (unsigned long long)(it_end - it_begin) | ((__builtin_amdgcn_s_memrealtime() - ig_rt0) << 32);
      }
#endif
    };
    if constexpr (FG) run_loop(load_slices_fast);
    else run_loop(load_slices_generic);
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
  if constexpr (H2) {
    if (h2_bad) *p.range_flag = 1;
  }
  // ---- epilogue.  C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int rowbase = m0 + wm * (32 * TM);
  const int colbase = n0 + wn * (32 * TN);
  if (splitk > 1) {   // raw partial slab [ks][M][N]
    float* slab = ws + ((long long)bz * splitk + ks) * p.M * p.N;
    const bool inl = p.splitk_counters != nullptr;       // in-launch combine (below) or the two-launch form
    if (wk == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = colbase + j * 32 + l31;
        if (col >= p.N) continue;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < p.M) {
              float* d = slab + (long long)row * p.N + col;
              if (inl) __hip_atomic_store(d, acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_store ... sc1
              else *d = acc[i][j][r];
            }
          }
      }
    }
    if (!inl) return;    // two-launch form: igemm_reduce_kernel sums the slabs and runs the epilogue
    // In-launch combine: the LAST of the tile's `splitk` workgroups to arrive sums all slabs (fixed order k = 0..splitk-1,
    // so the result does not depend on which one that is) and runs the ordinary epilogue below -- no reduce launch, no
    // second pass over the output.  Hand-off per MI355X_MICROARCH.md (per-XCD L2s are not coherent, a CU's L1 is never
    // refreshed): the slabs are stored write-through (sc1) and read back with sc1 loads -- EVERY store and EVERY load of the
    // handed-off bytes -- every storing wave drains its stores (vmcnt(0)), then a workgroup barrier, then ONE lane takes an
    // agent-scope ticket; the wave that drew the last ticket loads behind a workgroup barrier.  No agent-scope fence:
    // a release fence (buffer_wbl2) writes back EVERYTHING dirty in the XCD's L2 -- with ~100 MB of activations in flight
    // that made the fused form 1.5-2x slower than the reduce launch it replaces (measured; kept out).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* flag = reinterpret_cast<int*>(smem);             // the staging buffer is free now (all MFMA reads are done)
    if (tid == 0) {
      int* cnt = p.splitk_counters + (long long)bz * tiles_m * tiles_n + bid;
      const int ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == splitk - 1;
      if (last) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zeroed for the next launch
      *flag = last;
    }
    __syncthreads();
    if (*flag == 0 || wk != 0) return;
    // Non-atomic sc1 buffer loads (aux bit 4), so the compiler keeps a whole batch in flight: up to 4 slabs x 16 rows
    // per 32x32 tile before the first add.  (One relaxed atomic load per element serialised on its own s_waitcnt:
    // 16 slabs x ~2 us each, which is what made the first version of this path 1.5-2x slower than the reduce launch.)
    const float* slab0 = ws + (long long)bz * splitk * p.M * p.N;
    const unsigned slab_bytes = (unsigned)((long long)p.M * p.N * 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(slab0), 0,
                                                                         (int)(slab_bytes * (unsigned)splitk), 0x00020000);
    constexpr int SC1 = 1 << 4;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = min(colbase + j * 32 + l31, p.N - 1);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        int off[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.M - 1);
          off[r] = (row * p.N + col) * 4;
          acc[i][j][r] = 0.f;
        }
        for (int k0 = 0; k0 < splitk; k0 += 4) {
          float t[4][16];
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            // slabs past the last one: an offset beyond num_records reads as 0
            const unsigned so = (k0 + kk < splitk) ? (unsigned)(k0 + kk) * slab_bytes : 0xFFFFFFF0u;
#pragma unroll
            for (int r = 0; r < 16; ++r)
              t[kk][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off[r], (int)so, SC1));
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk)      // fixed order k = 0 .. splitk-1
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += t[kk][r];
        }
      }
    }
  } else if (wk != 0) {
    return;
  }
  {
    // (wave-uniform; LDMK_IG_LEAN=0 -> bit 1 of nfast: the general form everywhere, A/B)
    if (!(nfast & 2) && splitk == 1 && p.a_tf != LDMK_TF_LAYERNORM_FOLDED && p.epi != LDMK_EPI_GEGLU && rowbase + 32 * TM <= p.M &&
        colbase + 32 * TN <= p.N && !(p.batch_vec && p.rows_per_sample % 32 != 0)) {
      if (p.batch_vec && p.residual) ig_lean_epilogue<TM, TN, 4>(p, acc, rowbase, colbase, bz, l31, half);
      else if (p.batch_vec) ig_lean_epilogue<TM, TN, 1>(p, acc, rowbase, colbase, bz, l31, half);
      else if (p.residual) ig_lean_epilogue<TM, TN, 2>(p, acc, rowbase, colbase, bz, l31, half);
      else ig_lean_epilogue<TM, TN, 3>(p, acc, rowbase, colbase, bz, l31, half);
      return;
    }
  }
  float* __restrict__ outp = p.out + (long long)bz * p.out_bstride;
  const float* resp = p.residual ? p.residual + (long long)bz * p.out_bstride : nullptr;
  const float alpha = p.alpha;
  const bool lnf = p.a_tf == LDMK_TF_LAYERNORM_FOLDED;
  const float2* __restrict__ stats2 = reinterpret_cast<const float2*>(p.row_stats);
  if (p.epi == LDMK_EPI_GEGLU) {
    if constexpr (TN % 2 == 0) {
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const int cv = colbase + j * 32 + l31;        // packed value column
        const int cg = cv + 32;                       // packed gate column
        if (cv >= p.N) continue;
        const int oc = ((colbase + j * 32) >> 1) + l31;
        const float bv = p.bias ? p.bias[cv] : 0.f, bg = p.bias ? p.bias[cg] : 0.f;
        const float csv = lnf ? p.ln_colsum[cv] : 0.f, csg = lnf ? p.ln_colsum[cg] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          float2 st[16];
          if (lnf) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
              st[r] = stats2[min(rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.M - 1)];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < p.M) {
              float v = acc[i][j][r] * alpha, g = acc[i][j + 1][r] * alpha;
              if (lnf) {      // same arithmetic as rgemm.hip and igemm_reduce_kernel
                v = fmaf(-st[r].x, csv, v) * st[r].y;
                g = fmaf(-st[r].x, csg, g) * st[r].y;
              }
              v += bv;
              g += bg;
              const float ge = gelu_erf_f(g);                                            // exact (erf) GELU
              outp[(long long)row * p.ldc + oc] = v * ge;
            }
          }
        }
      }
    }
    return;
  }
  // element offsets are 32-bit (the host checks M * ldc < 2^31): one add per address; the per-sample vector is looked up
  // once per 32-row tile when a tile cannot straddle two samples (rows_per_sample % 32 == 0: every UNet / VQGAN level)
  // instead of one integer division per output element
  const bool tile_in_sample = p.rows_per_sample % 32 == 0;
  int smp[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) smp[i] = p.batch_vec ? min(rowbase + i * 32, p.M - 1) / p.rows_per_sample : 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = colbase + j * 32 + l31;
    if (col >= p.N) continue;
    const float bv = p.bias ? p.bias[col] : 0.f;
    const float cs = lnf ? p.ln_colsum[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float vals[16];
      const int r0 = rowbase + i * 32 + 4 * half;
      if (lnf) {
        float2 st[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = stats2[min(r0 + (r & 3) + 8 * (r >> 2), p.M - 1)];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = fmaf(-st[r].x, cs, acc[i][j][r] * alpha) * st[r].y;
      }
      const unsigned obase = (unsigned)r0 * (unsigned)p.ldc + (unsigned)col;
      const float vec = (p.batch_vec && tile_in_sample) ? p.batch_vec[(long long)smp[i] * p.batch_vec_ld + col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = (r & 3) + 8 * (r >> 2);
        float v = 0.f;
        if (r0 + dr < p.M) {
          v = (lnf ? acc[i][j][r] : acc[i][j][r] * alpha) + bv;
          if (p.batch_vec) v += tile_in_sample ? vec : p.batch_vec[(long long)((r0 + dr) / p.rows_per_sample) * p.batch_vec_ld + col];
          const unsigned o = obase + (unsigned)(dr * p.ldc);
          if (resp) v += resp[o];
          outp[o] = v;
        }
        vals[r] = v;
      }
      if (p.stats_out && rowbase + i * 32 < p.M) {
        // GroupNorm partial record of this 32-row tile x column (same record as gn_partial_kernel):
        // the 32 rows of the tile sit in 16 registers x 2 half-waves of the lane pair (l31, l31+32)
        const float shift = __shfl(vals[0], l31, 64);      // row 0 of the tile
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = vals[r] - shift;
          sm += d;
          sq = fmaf(d, d, sq);
        }
        sm += __shfl_xor(sm, 32, 64);
        sq += __shfl_xor(sq, 32, 64);
        if (half == 0) {
          float* d = p.stats_out + ((long long)((rowbase + i * 32) >> 5) * p.N + col) * 3;
          d[0] = shift; d[1] = sm; d[2] = sq;
        }
      }
    }
  }
}

// sum of the split-K slabs in a fixed order + the igemm epilogue (bias, per-sample vector, residual)
__global__ __launch_bounds__(256) void igemm_reduce_kernel(const ldmk_igemm_args p, const int splitk,
                                                           const float* __restrict__ ws) {
  const int n4 = p.N / 4;
  const long long total = (long long)p.M * n4;
  const int bz = blockIdx.z;
  const float* slab0 = ws + (long long)bz * splitk * p.M * p.N;
  float* outp = p.out + (long long)bz * p.out_bstride;
  const float* resp = p.residual ? p.residual + (long long)bz * p.out_bstride : nullptr;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int row = (int)(i / n4);
    const int col = (int)(i - (long long)row * n4) * 4;
    float4 s = *reinterpret_cast<const float4*>(slab0 + (long long)row * p.N + col);
    for (int k = 1; k < splitk; ++k) {
      const float4 t = *reinterpret_cast<const float4*>(slab0 + ((long long)k * p.M + row) * p.N + col);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    s.x *= p.alpha; s.y *= p.alpha; s.z *= p.alpha; s.w *= p.alpha;
    if (p.a_tf == LDMK_TF_LAYERNORM_FOLDED) {
      const float2 st = reinterpret_cast<const float2*>(p.row_stats)[row];
      const float4 c = *reinterpret_cast<const float4*>(p.ln_colsum + col);
      s.x = fmaf(-st.x, c.x, s.x) * st.y; s.y = fmaf(-st.x, c.y, s.y) * st.y;
      s.z = fmaf(-st.x, c.z, s.z) * st.y; s.w = fmaf(-st.x, c.w, s.w) * st.y;
    }
    if (p.bias) {
      const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (p.batch_vec) {
      const float4 b = *reinterpret_cast<const float4*>(p.batch_vec + (long long)(row / p.rows_per_sample) * p.batch_vec_ld + col);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    const long long o = (long long)row * p.ldc + col;
    if (resp) {
      const float4 b = *reinterpret_cast<const float4*>(resp + o);
      s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    *reinterpret_cast<float4*>(outp + o) = s;
  }
}

// split-K reduce + epilogue that also emits the GroupNorm partial records: one workgroup per
// (32-row tile, 64-column chunk); the reduced tile is kept in LDS for the per-column statistics.
__global__ __launch_bounds__(256) void igemm_reduce_stats_kernel(const ldmk_igemm_args p, const int splitk,
                                                                 const float* __restrict__ ws) {
  __shared__ float tile[32][65];
  const int rt = blockIdx.x, c0 = blockIdx.y * 64;
  float* outp = p.out;
  const float* resp = p.residual;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int idx = threadIdx.x + 256 * pass;        // 512 float4 = 32 rows x 16
    const int r = idx >> 4, cl = (idx & 15) * 4;
    const int row = rt * 32 + r, col = c0 + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (col < p.N) {
      s = *reinterpret_cast<const float4*>(ws + (long long)row * p.N + col);
      for (int k = 1; k < splitk; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(ws + ((long long)k * p.M + row) * p.N + col);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      s.x *= p.alpha; s.y *= p.alpha; s.z *= p.alpha; s.w *= p.alpha;
      if (p.bias) {
        const float4 b = *reinterpret_cast<const float4*>(p.bias + col);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      }
      if (p.batch_vec) {
        const float4 b = *reinterpret_cast<const float4*>(p.batch_vec + (long long)(row / p.rows_per_sample) * p.batch_vec_ld + col);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      }
      const long long o = (long long)row * p.ldc + col;
      if (resp) {
        const float4 b = *reinterpret_cast<const float4*>(resp + o);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
      }
      *reinterpret_cast<float4*>(outp + o) = s;
    }
    tile[r][cl] = s.x; tile[r][cl + 1] = s.y; tile[r][cl + 2] = s.z; tile[r][cl + 3] = s.w;
  }
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < p.N) {
    const int c = threadIdx.x;
    const float shift = tile[0][c];
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const float d = tile[r][c] - shift;
      sm += d;
      sq = fmaf(d, d, sq);
    }
    float* d = p.stats_out + ((long long)rt * p.N + c0 + c) * 3;
    d[0] = shift; d[1] = sm; d[2] = sq;
  }
}

// W[K][ldb] fp32 -> the three bf16 images [3][N][ld_out] of its exact split, K-contiguous (LDMK_COMPUTE_BF16X3)
template <int NIMG>
__global__ __launch_bounds__(256) void pack_wsplit_kernel(const float* __restrict__ w, int K, int N, int ldb, long long w_bstride,
                                                          __bf16* __restrict__ out, int ld_out) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const float* wb = w + (long long)blockIdx.z * w_bstride;
  __bf16* ob = out + (long long)blockIdx.z * NIMG * N * ld_out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = (threadIdx.x >> 5) + 8 * r, n = threadIdx.x & 31;
    tile[k][n] = (k0 + k < K && n0 + n < N) ? wb[(long long)(k0 + k) * ldb + n0 + n] : 0.f;
  }
  __syncthreads();
  const int n = threadIdx.x >> 3, kq = (threadIdx.x & 7) * 4;
  if (n0 + n < N && k0 + kq < ld_out) {
    const float4 v = make_float4(tile[kq][n], tile[kq + 1][n], tile[kq + 2][n], tile[kq + 3][n]);
    bf16x4 h, m, l;
    split3(v, h, m, l);
    __bf16* d = ob + (long long)(n0 + n) * ld_out + k0 + kq;
    *reinterpret_cast<bf16x4*>(d) = h;                       // (NIMG = 1: the round-to-nearest-even bf16 image alone)
    if constexpr (NIMG == 3) {
      *reinterpret_cast<bf16x4*>(d + (long long)N * ld_out) = m;
      *reinterpret_cast<bf16x4*>(d + 2LL * N * ld_out) = l;
    }
  }
}

// W[K][ldb] fp32 -> the two fp16 images [2][N][ld_out] of 2^scale_exp W, K-contiguous (LDMK_COMPUTE_F16X2)
__global__ __launch_bounds__(256) void pack_wsplit_h2_kernel(const float* __restrict__ w, int K, int N, int ldb, long long w_bstride, float scale,
                                                             _Float16* __restrict__ out, int ld_out) {
  __shared__ float tile[32][33];
  const int k0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  const float* wb = w + (long long)blockIdx.z * w_bstride;
  _Float16* ob = out + (long long)blockIdx.z * 2 * N * ld_out;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int k = (threadIdx.x >> 5) + 8 * r, n = threadIdx.x & 31;
    tile[k][n] = (k0 + k < K && n0 + n < N) ? wb[(long long)(k0 + k) * ldb + n0 + n] * scale : 0.f;
  }
  __syncthreads();
  const int n = threadIdx.x >> 3, kq = (threadIdx.x & 7) * 4;
  if (n0 + n < N && k0 + kq < ld_out) {
    f16x4 h, l;
    split2h(make_float4(tile[kq][n], tile[kq + 1][n], tile[kq + 2][n], tile[kq + 3][n]), h, l);
    _Float16* d = ob + (long long)(n0 + n) * ld_out + k0 + kq;
    *reinterpret_cast<f16x4*>(d) = h;
    *reinterpret_cast<f16x4*>(d + (long long)N * ld_out) = l;
  }
}

struct TileCfg { int bm, bn, ns; bool even_tn; float eff; };
// eff: measured sustained fraction of the f32 MFMA peak on long-K problems (profiles/r01_layers*_v2.txt, refined with
// the tools/autotune.py sweeps: the 128x128 tile reaches 0.76 on the VQGAN decoder convolutions).  Shapes the
// sweeps covered never get here -- engine.tuned_plan() answers from dsml_thesis_amd/igemm_plans.json first.
static const TileCfg kCfg[] = {
    {128, 128, 1, true, 0.70f},   // 1: <2,2,2,2,1,1>
    {64, 128, 2, true, 0.58f},    // 2: <1,2,2,2,1,2>
    {64, 64, 4, true, 0.45f},     // 3: <2,2,1,1,4,1>
    {64, 64, 2, false, 0.55f},    // 4: <1,1,2,2,1,2>
    {128, 160, 1, false, 0.76f},  // 5: <1,5,4,1,1,1>
    {64, 160, 2, false, 0.62f},   // 6: <1,5,2,1,2,1>
};
constexpr int kNumCfg = sizeof(kCfg) / sizeof(kCfg[0]);

template <int TM, int TN, int WM, int WN, int WK, int KS, bool DB, bool BT, int BF = 0>
static size_t cfg_lds_bytes() {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN, KC = 32 * WK * KS;
  constexpr int ASTR = BM + 1, BSTR = BN + (BT ? 1 : 0);
  if (BF == 3) return (size_t)3 * (BM + BN) * (KC + 8) * 2;
  if (BF == 4) return (size_t)2 * (BM + BN) * (KC + 8) * 2;
  if (BF == 2) return (size_t)(BM + BN) * (KC + 8) * 2;
  if (BF) return BT ? (size_t)(BM + BN) * (KC + 8) * 2 : (size_t)BM * (KC + 8) * 2 + (size_t)KC * BN * 4;
  size_t stage = (size_t)KC * (ASTR + BSTR) * sizeof(float) * (DB ? 2 : 1);
  size_t red = WK > 1 ? (size_t)(WK - 1) * WM * WN * TM * TN * 16 * 64 * sizeof(float) : 0;
  return stage > red ? stage : red;
}

template <int TM, int TN, int WM, int WN, int WK, int KS, bool DB, bool BT, int BF = 0, bool FG = true, bool ASP = false>
static bool& cfg_attr_done() {
  static bool done = false;
  return done;
}

template <int TM, int TN, int WM, int WN, int WK, int KS, bool DB, bool BT, int BF = 0, bool FG = true, bool ASP = false>
static void cfg_set_attr() {
  bool& done = cfg_attr_done<TM, TN, WM, WN, WK, KS, DB, BT, BF, FG, ASP>();
  if (!done) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_kernel<TM, TN, WM, WN, WK, KS, DB, BT, BF, FG, ASP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg_lds_bytes<TM, TN, WM, WN, WK, KS, DB, BT, BF>());
    done = true;
  }
}

// the fast gather's preconditions (32-bit byte offsets into each source, no upsampling index math)
static bool igemm_fast_gather_ok(const ldmk_igemm_args& a) {
  const long long samples = ((long long)a.M + a.rows_per_sample - 1) / a.rows_per_sample;
  const long long rows = a.a_mode == LDMK_A_CONV3X3 ? samples * a.in_h * a.in_w : (long long)a.M;
  const long long wb = (a.b_trans ? (long long)a.N : (long long)a.K) * a.ldb * 4;
  const bool ups_ok = a.upsample == 0 || (a.upsample == 1 && a.c1 == 0 && a.a_mode == LDMK_A_CONV3X3);
  return ups_ok && rows * a.c0 * 4 < (1LL << 32) && rows * a.c1 * 4 < (1LL << 32) && wb < (1LL << 32) && (a.batch <= 1 || a.a1 == nullptr);   // (a batch offsets a0 and w only)
}

// the second launch of a split-K GEMM: slabs summed in slab order + the epilogue (also used by the slab GEMM, sgemm.hip)
int launch_splitk_reduce(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  if (a.stats_out) {
    hipLaunchKernelGGL(igemm_reduce_stats_kernel, dim3(a.M / 32, (a.N + 63) / 64), dim3(256), 0, st, a, splitk, ws);
  } else {
    long long total = (long long)a.M * (a.N / 4);
    int g = (int)((total + 255) / 256);
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(igemm_reduce_kernel, dim3(g, 1, a.batch > 1 ? a.batch : 1), dim3(256), 0, st, a, splitk, ws);
  }
  return check_launch("ldmk_igemm(reduce)");
}

template <int TM, int TN, int WM, int WN, int WK, int KS, bool DB, bool BT, int BF, bool FG, bool ASP = false>
static int launch_cfg_g(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st);

template <int TM, int TN, int WM, int WN, int WK, int KS, bool DB, bool BT, int BF = 0>
static int launch_cfg(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  return igemm_fast_gather_ok(a) ? launch_cfg_g<TM, TN, WM, WN, WK, KS, DB, BT, BF, true>(a, splitk, ws, st)
                                 : launch_cfg_g<TM, TN, WM, WN, WK, KS, DB, BT, BF, false>(a, splitk, ws, st);
}

template <int TM, int TN, int WM, int WN, int WK, int KS, bool DB, bool BT, int BF, bool FG, bool ASP>
static int launch_cfg_g(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  constexpr int BM = 32 * TM * WM, BN = 32 * TN * WN;
  size_t lds = cfg_lds_bytes<TM, TN, WM, WN, WK, KS, DB, BT, BF>();
  int tiles = ((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN);
  dim3 grid(tiles, splitk, a.batch > 1 ? a.batch : 1);
  auto k = igemm_kernel<TM, TN, WM, WN, WK, KS, DB, BT, BF, FG, ASP>;
  cfg_set_attr<TM, TN, WM, WN, WK, KS, DB, BT, BF, FG, ASP>();
  static const int nfast_env = [] { const char* e = getenv("LDMK_IG_NFAST"); return e ? atoi(e) : 1; }();
  // tall problems only (M >> N: activations x a weight matrix); weight-gradient-like or b_trans shapes keep the old order
  static const int lean_env = [] { const char* e = getenv("LDMK_IG_LEAN"); return e ? atoi(e) : 1; }();      // (0: the general epilogue everywhere, A/B)
  const int nfast = (nfast_env && !a.b_trans && (a.N + BN - 1) / BN > 1 && (a.M + BM - 1) / BM >= 8 ? 1 : 0) | (lean_env ? 0 : 2);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, st, a, splitk, ws, nfast);
  if (splitk > 1 && !a.splitk_counters && !a.raw_slabs) return launch_splitk_reduce(a, splitk, ws, st);
  // (splitk_counters: the last-arriving workgroup of each tile combined the slabs and ran the epilogue inside the launch;
  //  raw_slabs: the consumer sums the slabs)
  return check_launch("ldmk_igemm");
}

// Tile shape + cross-workgroup K split for a problem size: maximise (tile efficiency) x (useful columns)
// x (occupancy of 512 workgroup slots = 2 per CU, counting whole rounds); splitting K costs one extra
// pass over the partial slabs.
static void plan(const ldmk_igemm_args& a, int* cfg_out, int* splitk_out, long long ws_elems) {
  const long long b = a.batch > 1 ? a.batch : 1;
  const bool geglu = a.epi == LDMK_EPI_GEGLU;
  const int nkc = a.K / 32;
  float best = -1.f;
  int best_cfg = 3, best_sk = 1;
  static const int sks[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (int c = 0; c < kNumCfg; ++c) {
    if (geglu && !kCfg[c].even_tn) continue;
    const long long tm = (a.M + kCfg[c].bm - 1) / kCfg[c].bm, tn = (a.N + kCfg[c].bn - 1) / kCfg[c].bn;
    const long long blocks = b * tm * tn;
    const float col_use = (float)a.N / (float)(tn * kCfg[c].bn);
    const float row_use = (float)a.M / (float)(tm * kCfg[c].bm);
    const int iters = (nkc + kCfg[c].ns - 1) / kCfg[c].ns;
    for (int si = 0; si < 8; ++si) {
      const int sk = sks[si];
      if (sk > 1 && (geglu || ws_elems <= 0 || iters / sk < 2 || b * sk * (long long)a.M * a.N > ws_elems)) continue;
      const long long w = blocks * sk;
      const long long rounds = (w + 511) / 512;
      const float occ = (float)w / (float)(rounds * 512);
      // short K: the per-workgroup prologue/epilogue weighs more on big tiles
      const float kpen = (float)(iters / sk) / (float)(iters / sk + 2);
      const float score = kCfg[c].eff * col_use * row_use * occ * kpen * (sk > 1 ? 0.93f : 1.f);
      if (score > best) { best = score; best_cfg = c; best_sk = sk; }
    }
  }
  *cfg_out = best_cfg + 1;
  *splitk_out = best_sk;
}

// bf16 matrix-core compute (args.compute = LDMK_COMPUTE_BF16): the WK = 1 tile shapes
template <bool BT>
static int dispatch_bf16(const ldmk_igemm_args& a, int cfg, int splitk, float* ws, hipStream_t st) {
  const bool geglu = a.epi == LDMK_EPI_GEGLU;
  if (cfg == 3) cfg = 4;                       // 64x64 with K split over the waves -> 64x64, 64 k per stage
  if (cfg == 6) cfg = geglu ? 2 : 5;           // 64x160 with K split over wave pairs -> 128x160
  if (geglu && !kCfg[cfg - 1].even_tn) cfg = 1;
  switch (cfg) {
    case 1: return launch_cfg<2, 2, 2, 2, 1, 1, false, BT, 1>(a, splitk, ws, st);
    case 2: return launch_cfg<1, 2, 2, 2, 1, 2, false, BT, 1>(a, splitk, ws, st);
    case 4: return launch_cfg<1, 1, 2, 2, 1, 2, false, BT, 1>(a, splitk, ws, st);
    default: return launch_cfg<1, 5, 4, 1, 1, 1, false, BT, 1>(a, splitk, ws, st);
  }
}

// fp32-accurate three-way bf16 split (args.compute = LDMK_COMPUTE_BF16X3): the WK = 1 tile shapes, pre-split weights
static int dispatch_x3(const ldmk_igemm_args& a, int cfg, int splitk, float* ws, hipStream_t st) {
  const bool geglu = a.epi == LDMK_EPI_GEGLU;
  if (cfg == 3) cfg = 4;
  if (cfg == 6) cfg = geglu ? 2 : 5;
  if (geglu && !kCfg[cfg - 1].even_tn) cfg = 1;
  if (a.a_split) {                             // the A operand pre-split as well (ldmk_ln_stats_split): copies only
    switch (cfg) {
      case 1: return launch_cfg_g<2, 2, 2, 2, 1, 1, false, false, 3, true, true>(a, splitk, ws, st);
      case 2: return launch_cfg_g<1, 2, 2, 2, 1, 2, false, false, 3, true, true>(a, splitk, ws, st);
      case 4: return launch_cfg_g<1, 1, 2, 2, 1, 2, false, false, 3, true, true>(a, splitk, ws, st);
      default: return launch_cfg_g<1, 5, 4, 1, 1, 1, false, false, 3, true, true>(a, splitk, ws, st);
    }
  }
  switch (cfg) {
    case 1: return launch_cfg_g<2, 2, 2, 2, 1, 1, false, false, 3, true>(a, splitk, ws, st);
    case 2: return launch_cfg_g<1, 2, 2, 2, 1, 2, false, false, 3, true>(a, splitk, ws, st);
    case 4: return launch_cfg_g<1, 1, 2, 2, 1, 2, false, false, 3, true>(a, splitk, ws, st);
    default: return launch_cfg_g<1, 5, 4, 1, 1, 1, false, false, 3, true>(a, splitk, ws, st);
  }
}

// fp32-accurate two-way fp16 split (args.compute = LDMK_COMPUTE_F16X2): the same tile shapes, three matrix instructions per product
static int dispatch_h2(const ldmk_igemm_args& a, int cfg, int splitk, float* ws, hipStream_t st) {
  const bool geglu = a.epi == LDMK_EPI_GEGLU;
  if (cfg == 3) cfg = 4;
  if (cfg == 6) cfg = geglu ? 2 : 5;
  if (geglu && !kCfg[cfg - 1].even_tn) cfg = 1;
  switch (cfg) {
    case 1: return launch_cfg_g<2, 2, 2, 2, 1, 1, false, false, 4, true>(a, splitk, ws, st);
    case 2: return launch_cfg_g<1, 2, 2, 2, 1, 2, false, false, 4, true>(a, splitk, ws, st);
    case 4: return launch_cfg_g<1, 1, 2, 2, 1, 2, false, false, 4, true>(a, splitk, ws, st);
    default: return launch_cfg_g<1, 5, 4, 1, 1, 1, false, false, 4, true>(a, splitk, ws, st);
  }
}

// bf16 compute with the weights pre-packed (args.w_split = one bf16 image, ldmk_pack_wbf16t): the training step's forward GEMMs
static int dispatch_bf16_packed(const ldmk_igemm_args& a, int cfg, int splitk, float* ws, hipStream_t st) {
  const bool geglu = a.epi == LDMK_EPI_GEGLU;
  if (cfg == 3) cfg = 4;
  if (cfg == 6) cfg = geglu ? 2 : 5;
  if (geglu && !kCfg[cfg - 1].even_tn) cfg = 1;
  switch (cfg) {
    case 1: return launch_cfg_g<2, 2, 2, 2, 1, 1, false, false, 2, true>(a, splitk, ws, st);
    case 2: return launch_cfg_g<1, 2, 2, 2, 1, 2, false, false, 2, true>(a, splitk, ws, st);
    case 4: return launch_cfg_g<1, 1, 2, 2, 1, 2, false, false, 2, true>(a, splitk, ws, st);
    default: return launch_cfg_g<1, 5, 4, 1, 1, 1, false, false, 2, true>(a, splitk, ws, st);
  }
}

template <bool BT>
static int dispatch(const ldmk_igemm_args& a, int cfg, int splitk, float* ws, hipStream_t st) {
  if (a.compute == LDMK_COMPUTE_BF16 && a.w_split && !BT && igemm_fast_gather_ok(a)) return dispatch_bf16_packed(a, cfg, splitk, ws, st);
  if (a.compute == LDMK_COMPUTE_BF16) return dispatch_bf16<BT>(a, cfg, splitk, ws, st);
  if (a.compute == LDMK_COMPUTE_BF16X3) return dispatch_x3(a, cfg, splitk, ws, st);
  if (a.compute == LDMK_COMPUTE_F16X2) return dispatch_h2(a, cfg, splitk, ws, st);
  if (a.epi == LDMK_EPI_GEGLU && !kCfg[cfg - 1].even_tn) cfg = 3;   // GEGLU needs (value, gate) tile pairs
  switch (cfg) {
    case 1: return launch_cfg<2, 2, 2, 2, 1, 1, false, BT>(a, splitk, ws, st);   // 128x128
    case 2: return launch_cfg<1, 2, 2, 2, 1, 2, false, BT>(a, splitk, ws, st);    // 64x128, 64 k per stage
    case 3: return launch_cfg<2, 2, 1, 1, 4, 1, false, BT>(a, splitk, ws, st);   // 64x64, K split over the 4 waves
    case 4: return launch_cfg<1, 1, 2, 2, 1, 2, false, BT>(a, splitk, ws, st);    // 64x64, 64 k per stage
    case 5: return launch_cfg<1, 5, 4, 1, 1, 1, false, BT>(a, splitk, ws, st);    // 128x160
    default: return launch_cfg<1, 5, 2, 1, 2, 1, false, BT>(a, splitk, ws, st);  // 64x160, K split over wave pairs
  }
}

template <bool BT, bool FG>
static void set_all_attrs() {
  cfg_set_attr<2, 2, 2, 2, 1, 1, false, BT, false, FG>();
  cfg_set_attr<1, 2, 2, 2, 1, 2, false, BT, false, FG>();
  cfg_set_attr<2, 2, 1, 1, 4, 1, false, BT, false, FG>();
  cfg_set_attr<1, 1, 2, 2, 1, 2, false, BT, false, FG>();
  cfg_set_attr<1, 5, 4, 1, 1, 1, false, BT, false, FG>();
  cfg_set_attr<1, 5, 2, 1, 2, 1, false, BT, false, FG>();
}
void igemm_init_attributes() {
  set_all_attrs<false, true>();
  set_all_attrs<true, true>();
  set_all_attrs<false, false>();
  set_all_attrs<true, false>();
}

// the wave-autonomous row GEMM (rgemm.hip): tile_cfg kNumCfg+1 .. kNumCfg+6
const char* rgemm_unsupported(const ldmk_igemm_args& a, int rcfg);
int rgemm_dispatch(const ldmk_igemm_args& a, int rcfg, hipStream_t st);
constexpr int kNumRCfg = 6;
// the slab GEMM for small row counts (sgemm.hip): tile_cfg kNumCfg+kNumRCfg+1 .. +8
const char* sgemm_unsupported(const ldmk_igemm_args& a, int scfg, int splitk);
int sgemm_dispatch(const ldmk_igemm_args& a, int scfg, int splitk, float* ws, hipStream_t st);
constexpr int kNumSCfg = 8;
// the warp-specialised bf16x3 tiles (igemm_ws.hip): tile_cfg kNumCfg+kNumRCfg+kNumSCfg+1 .. +2 (256x160, 256x128)
const char* igemm_ws_unsupported(const ldmk_igemm_args& a, int wcfg, int splitk);
int igemm_ws_dispatch(const ldmk_igemm_args& a, int wcfg, int splitk, float* ws, hipStream_t st);
constexpr int kNumWCfg = 2;
// the pre-split bf16x3 tiles (igemm_ps.hip): tile_cfg kNumCfg+kNumRCfg+kNumSCfg+kNumWCfg+1 .. +6
const char* igemm_ps_unsupported(const ldmk_igemm_args& a, int pcfg, int splitk);
int igemm_ps_dispatch(const ldmk_igemm_args& a, int pcfg, int splitk, float* ws, hipStream_t st);
constexpr int kNumPCfg = 11;

}  // namespace ldmk

extern "C" int ldmk_pack_wsplit(const float* w, int K, int N, int ldb, int batch, long long w_bstride, void* out, int ld_out,
                                void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(w && out && K > 0 && N > 0 && ldb >= N && batch >= 1, "ldmk_pack_wsplit: bad args");
  LDMK_REQUIRE(ld_out >= K && ld_out % 8 == 0, "ldmk_pack_wsplit: ld_out=%d must be >= K=%d and a multiple of 8", ld_out, K);
  hipLaunchKernelGGL(ldmk::pack_wsplit_kernel<3>, dim3((ld_out + 31) / 32, (N + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream, w, K,
                     N, ldb, w_bstride, reinterpret_cast<__bf16*>(out), ld_out);
  return ldmk::check_launch("ldmk_pack_wsplit");
}

extern "C" int ldmk_pack_wsplit_h2(const float* w, int K, int N, int ldb, int batch, long long w_bstride, int scale_exp, void* out, int ld_out,
                                   void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(w && out && K > 0 && N > 0 && ldb >= N && batch >= 1, "ldmk_pack_wsplit_h2: bad args");
  LDMK_REQUIRE(ld_out >= K && ld_out % 8 == 0, "ldmk_pack_wsplit_h2: ld_out=%d must be >= K=%d and a multiple of 8", ld_out, K);
  LDMK_REQUIRE(scale_exp >= -60 && scale_exp <= 60, "ldmk_pack_wsplit_h2: scale_exp=%d outside [-60, 60]", scale_exp);
  hipLaunchKernelGGL(ldmk::pack_wsplit_h2_kernel, dim3((ld_out + 31) / 32, (N + 31) / 32, batch), dim3(256), 0, (hipStream_t)stream, w, K, N, ldb,
                     w_bstride, ldexpf(1.f, scale_exp), reinterpret_cast<_Float16*>(out), ld_out);
  return ldmk::check_launch("ldmk_pack_wsplit_h2");
}

extern "C" int ldmk_pack_wbf16t(const float* w, int K, int N, int ldb, void* out, int ld_out, void* stream) {
  LDMK_ENTER();
  LDMK_REQUIRE(w && out && K > 0 && N > 0 && ldb >= N, "ldmk_pack_wbf16t: bad args");
  LDMK_REQUIRE(ld_out >= K && ld_out % 8 == 0, "ldmk_pack_wbf16t: ld_out=%d must be >= K=%d and a multiple of 8", ld_out, K);
  hipLaunchKernelGGL(ldmk::pack_wsplit_kernel<1>, dim3((ld_out + 31) / 32, (N + 31) / 32, 1), dim3(256), 0, (hipStream_t)stream, w, K, N,
                     ldb, 0LL, reinterpret_cast<__bf16*>(out), ld_out);
  return ldmk::check_launch("ldmk_pack_wbf16t");
}

// test hook: force a tile configuration (0 = heuristic)
static int g_force_cfg = 0;
extern "C" void ldmk_igemm_force_config(int cfg) { g_force_cfg = cfg; }

extern "C" int ldmk_igemm_plan(const ldmk_igemm_args* args, int* tile_cfg, int* splitk) {
  if (!args || !tile_cfg || !splitk) return LDMK_EINVAL;
  ldmk::plan(*args, tile_cfg, splitk, args->splitk_ws ? args->splitk_ws_elems : 0);
  return LDMK_OK;
}

extern "C" long long ldmk_igemm_workspace_elems(const ldmk_igemm_args* args) {
  using namespace ldmk;
  LDMK_REQUIRE(args != nullptr, "ldmk_igemm_workspace_elems: null args");
  const ldmk_igemm_args& a = *args;
  LDMK_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "ldmk_igemm_workspace_elems: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  LDMK_REQUIRE(a.splitk >= 0 && a.splitk <= 64, "ldmk_igemm_workspace_elems: splitk=%d outside [0,64]", a.splitk);
  int cfg = a.tile_cfg, sk = a.splitk;
  if (sk == 0) {
    int c2 = 0;
    plan(a, &c2, &sk, 1LL << 50);          // what the planner would use with unlimited scratch
    if (cfg == 0) cfg = c2;
  }
  if (cfg > kNumCfg + kNumRCfg + kNumSCfg)                                  // warp-specialised bf16x3 tiles: split like the LDS-tiled ones
    return sk <= 1 || a.epi == LDMK_EPI_GEGLU ? 0 : (long long)(a.batch > 1 ? a.batch : 1) * sk * (long long)a.M * a.N;
  if (cfg > kNumCfg + kNumRCfg) {                                           // slab GEMM: GEGLU may split when the consumer reduces
    if (sk <= 1 || (a.epi == LDMK_EPI_GEGLU && !a.raw_slabs)) return 0;
    return (long long)sk * (long long)a.M * a.N;
  }
  if (cfg > kNumCfg || a.epi == LDMK_EPI_GEGLU || sk <= 1) return 0;      // row GEMM / GEGLU never split K
  return (long long)(a.batch > 1 ? a.batch : 1) * sk * (long long)a.M * a.N;
}

static int igemm_entry(const ldmk_igemm_args* args, void* stream, bool launch);
extern "C" int ldmk_igemm(const ldmk_igemm_args* args, void* stream) { return igemm_entry(args, stream, true); }
extern "C" int ldmk_igemm_check(const ldmk_igemm_args* args) { return igemm_entry(args, nullptr, false); }

static int igemm_entry(const ldmk_igemm_args* args, void* stream, bool launch) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(args != nullptr, "ldmk_igemm: null args");
  ldmk_igemm_args a = *args;
  LDMK_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "ldmk_igemm: empty problem M=%d N=%d K=%d", a.M, a.N, a.K);
  LDMK_REQUIRE(a.c0 > 0 && a.c0 % 32 == 0 && a.c1 % 32 == 0, "ldmk_igemm: channel counts must be multiples of 32 (c0=%d c1=%d)", a.c0, a.c1);
  LDMK_REQUIRE((a.c1 == 0) == (a.a1 == nullptr), "ldmk_igemm: a1/c1 mismatch");
  const int taps = a.a_mode == LDMK_A_CONV3X3 ? 9 : 1;
  const int skip_k = a.skip_a0 ? a.skip_c0 + a.skip_c1 : 0;       // fused 1x1 skip connection (slab GEMM only, checked there)
  LDMK_REQUIRE(a.K == taps * (a.c0 + a.c1) + skip_k, "ldmk_igemm: K=%d != taps*(c0+c1)+skip=%d", a.K, taps * (a.c0 + a.c1) + skip_k);
  LDMK_REQUIRE(!a.skip_a0 || (a.a_mode == LDMK_A_CONV3X3 && a.skip_c0 > 0 && a.skip_c0 % 8 == 0 && a.skip_c1 % 8 == 0 &&
                              (a.skip_c1 == 0) == (a.skip_a1 == nullptr) && (a.tile_cfg > kNumCfg + kNumRCfg || a.tile_cfg == 0)),
               "ldmk_igemm: the fused skip connection needs a 3x3 convolution on a slab-GEMM tile, skip channels in multiples of 8");
  LDMK_REQUIRE(a.N % 4 == 0 && a.ldb % 4 == 0, "ldmk_igemm: N and ldb must be multiples of 4");
  LDMK_REQUIRE(a.rows_per_sample > 0, "ldmk_igemm: rows_per_sample");
  LDMK_REQUIRE((long long)a.M * a.ldc < (1LL << 31) && a.ldc > 0, "ldmk_igemm: output exceeds 2^31 elements per batch item (32-bit epilogue offsets)");
  const long long samples = ((long long)a.M + a.rows_per_sample - 1) / a.rows_per_sample;
  if (a.a_mode == LDMK_A_CONV3X3) {
    LDMK_REQUIRE(a.in_h > 0 && a.in_w > 0 && a.out_h > 0 && a.out_w > 0 && a.stride >= 1, "ldmk_igemm: conv geometry");
    LDMK_REQUIRE(a.rows_per_sample == a.out_h * a.out_w, "ldmk_igemm: rows_per_sample != out_h*out_w");
    LDMK_REQUIRE(a.batch <= 1, "ldmk_igemm: batched conv unsupported");
    LDMK_REQUIRE(samples * a.in_h * a.in_w * (a.c0 > a.c1 ? a.c0 : a.c1) < (1LL << 31),
                 "ldmk_igemm: conv input exceeds 2^31 elements (32-bit gather indices)");
  } else {
    LDMK_REQUIRE((long long)a.M * (a.c0 > a.c1 ? a.c0 : a.c1) < (1LL << 31), "ldmk_igemm: A exceeds 2^31 elements per batch item");
  }
  if (a.a_tf == LDMK_TF_AFFINE || a.a_tf == LDMK_TF_AFFINE_SILU) LDMK_REQUIRE(a.tf_coef, "ldmk_igemm: tf_coef missing");
  if (a.a_tf == LDMK_TF_LAYERNORM) {
    LDMK_REQUIRE(a.row_stats && a.ln_gamma && a.ln_beta && a.a_mode == LDMK_A_ROWS, "ldmk_igemm: layernorm prologue args");
  }
  if (a.a_tf == LDMK_TF_LAYERNORM_FOLDED)
    LDMK_REQUIRE(a.row_stats && a.ln_colsum && a.a_mode == LDMK_A_ROWS && !a.stats_out && a.batch <= 1,
                 "ldmk_igemm: folded layernorm needs row_stats, ln_colsum (ldmk_fold_layernorm), rows mode, no stats_out, no batching");
  LDMK_REQUIRE(a.a_tf >= LDMK_TF_NONE && a.a_tf <= LDMK_TF_LAYERNORM_FOLDED, "ldmk_igemm: a_tf=%d", a.a_tf);
  if (a.epi == LDMK_EPI_GEGLU) LDMK_REQUIRE(a.N % 64 == 0 && !a.residual && !a.batch_vec, "ldmk_igemm: GEGLU needs N%%64==0 and no residual");
  if (a.stats_out)
    LDMK_REQUIRE(a.M % 32 == 0 && a.rows_per_sample % 32 == 0 && a.epi == LDMK_EPI_NONE && a.batch <= 1,
                 "ldmk_igemm: stats_out needs M%%32==0, rows_per_sample%%32==0, no GEGLU, no batching");
  LDMK_REQUIRE(a.tile_cfg >= 0 && a.tile_cfg <= kNumCfg + kNumRCfg + kNumSCfg + kNumWCfg + kNumPCfg, "ldmk_igemm: tile_cfg=%d outside [0,%d]",
               a.tile_cfg, kNumCfg + kNumRCfg + kNumSCfg + kNumWCfg + kNumPCfg);
  LDMK_REQUIRE(a.splitk >= 0 && a.splitk <= 64, "ldmk_igemm: splitk=%d outside [0,64]", a.splitk);
  LDMK_REQUIRE(a.compute == LDMK_COMPUTE_F32 || a.compute == LDMK_COMPUTE_BF16 || a.compute == LDMK_COMPUTE_BF16X3 || a.compute == LDMK_COMPUTE_F16X2,
               "ldmk_igemm: compute=%d", a.compute);
  const bool ps_tile = a.tile_cfg > kNumCfg + kNumRCfg + kNumSCfg + kNumWCfg;
  if (a.compute == LDMK_COMPUTE_BF16X3 && !ps_tile) {
    LDMK_REQUIRE(a.w_split && !a.b_trans && a.w_split_ld >= a.K && a.w_split_ld % 8 == 0,
                 "ldmk_igemm: LDMK_COMPUTE_BF16X3 needs w_split (ldmk_pack_wsplit), b_trans = 0, w_split_ld >= K and a multiple of 8");
    LDMK_REQUIRE(3LL * a.N * a.w_split_ld * 2 < (1LL << 32), "ldmk_igemm: w_split exceeds 4 GB per batch entry");
    LDMK_REQUIRE(igemm_fast_gather_ok(a), "ldmk_igemm: LDMK_COMPUTE_BF16X3 needs the fast gather (no zero-insertion, two-source "
                 "upsampling or operands beyond 4 GB)");
  }
  if (a.compute == LDMK_COMPUTE_F16X2) {
    LDMK_REQUIRE(a.tile_cfg <= kNumCfg || ps_tile, "ldmk_igemm: LDMK_COMPUTE_F16X2 runs on the LDS-tiled shapes (tile_cfg 0..6) and the pre-split "
                 "tiles (23..28, 31..33), not tile_cfg=%d", a.tile_cfg);
    LDMK_REQUIRE(!a.a_split && a.w_scale_exp >= -60 && a.w_scale_exp <= 60, "ldmk_igemm: LDMK_COMPUTE_F16X2: a_split is not taken; w_scale_exp=%d", a.w_scale_exp);
    if (!ps_tile) {
      LDMK_REQUIRE(a.w_split && !a.b_trans && a.w_split_ld >= a.K && a.w_split_ld % 8 == 0 && 2LL * a.N * a.w_split_ld * 2 < (1LL << 32),
                   "ldmk_igemm: LDMK_COMPUTE_F16X2 needs w_split (ldmk_pack_wsplit_h2), b_trans = 0, w_split_ld >= K and a multiple of 8, below 4 GB");
      LDMK_REQUIRE(a.range_flag != nullptr, "ldmk_igemm: LDMK_COMPUTE_F16X2 needs range_flag");
      LDMK_REQUIRE(igemm_fast_gather_ok(a), "ldmk_igemm: LDMK_COMPUTE_F16X2 needs the fast gather (no zero-insertion, two-source upsampling or "
                   "operands beyond 4 GB)");
    }
  }
  if (a.compute == LDMK_COMPUTE_BF16 && a.w_split)
    LDMK_REQUIRE(!a.b_trans && a.w_split_ld >= a.K && a.w_split_ld % 8 == 0 && (long long)a.N * a.w_split_ld * 2 < (1LL << 32) && a.batch <= 1,
                 "ldmk_igemm: LDMK_COMPUTE_BF16 with w_split (ldmk_pack_wbf16t) needs b_trans = 0, w_split_ld >= K and a multiple of 8, "
                 "no batching");
  if (a.a_split) {
    LDMK_REQUIRE(a.compute == LDMK_COMPUTE_BF16X3 && a.a_mode == LDMK_A_ROWS && a.c1 == 0 && a.batch <= 1 &&
                 (a.a_tf == LDMK_TF_NONE || a.a_tf == LDMK_TF_LAYERNORM_FOLDED) && a.tile_cfg <= kNumCfg,
                 "ldmk_igemm: a_split (pre-split A) needs LDMK_COMPUTE_BF16X3 on an LDS-tiled shape (tile_cfg 0..6), rows mode, one "
                 "source, no staging prologue, no batching");
    LDMK_REQUIRE(a.a_split_ld >= a.K && a.a_split_ld % 8 == 0 && 3LL * a.M * a.a_split_ld * 2 < (1LL << 32),
                 "ldmk_igemm: a_split_ld=%d must be >= K, a multiple of 8, and the three images below 4 GB", a.a_split_ld);
  }
  LDMK_REQUIRE(a.compute == LDMK_COMPUTE_F32 || a.tile_cfg <= kNumCfg || a.tile_cfg > kNumCfg + kNumRCfg + kNumSCfg,
               "ldmk_igemm: the row / slab GEMM tiles are fp32 only");
  if (a.alpha == 0.f) a.alpha = 1.f;
  if (a.compute == LDMK_COMPUTE_F16X2) a.alpha *= ldexpf(1.f, -(LDMK_F16X2_A_EXP + a.w_scale_exp));     // the operands' scales leave in the epilogue (exact)
  int cfg = 0, sk = 1;
  plan(a, &cfg, &sk, a.splitk_ws ? a.splitk_ws_elems : 0);
  if (a.tile_cfg > 0) cfg = a.tile_cfg;
  if (g_force_cfg > 0 && (g_force_cfg <= kNumCfg || !rgemm_unsupported(a, g_force_cfg - kNumCfg - 1))) cfg = g_force_cfg;
  LDMK_REQUIRE(!a.skip_a0 || cfg > kNumCfg + kNumRCfg, "ldmk_igemm: the fused skip connection runs on the slab GEMM only (tile_cfg 13..20)");
  if (cfg > kNumCfg + kNumRCfg + kNumSCfg + kNumWCfg) {      // pre-split bf16x3 tile: both operands in the PS layout, LDS-DMA staging
    const int pcfg = cfg - kNumCfg - kNumRCfg - kNumSCfg - kNumWCfg - 1;
    int psk = a.splitk > 0 ? a.splitk : 1;
    const char* why = igemm_ps_unsupported(a, pcfg, psk);
    LDMK_REQUIRE(why == nullptr, "ldmk_igemm: tile_cfg=%d splitk=%d (pre-split bf16x3 tile) cannot run this problem: %s", cfg, psk, why ? why : "");
    LDMK_REQUIRE(!a.raw_slabs || psk >= 2, "ldmk_igemm: raw_slabs needs a split-K plan");
    if (psk > 1) {
      const long long b = a.batch > 1 ? a.batch : 1;
      LDMK_REQUIRE_MEM(a.splitk_ws && b * psk * (long long)a.M * a.N <= a.splitk_ws_elems,
                       "ldmk_igemm: splitk=%d needs a workspace of %lld floats (ldmk_igemm_workspace_elems), %lld given", psk,
                       b * psk * (long long)a.M * a.N, a.splitk_ws ? a.splitk_ws_elems : 0LL);
    }
    if (!launch) return LDMK_OK;
    return igemm_ps_dispatch(a, pcfg, psk, a.splitk_ws, (hipStream_t)stream);
  }
  if (cfg > kNumCfg + kNumRCfg + kNumSCfg) {      // warp-specialised bf16x3 tile: producer / consumer waves, split-K over workgroups
    const int wcfg = cfg - kNumCfg - kNumRCfg - kNumSCfg - 1;
    int wsk = a.splitk > 0 ? a.splitk : 1;
    const char* why = igemm_ws_unsupported(a, wcfg, wsk);
    LDMK_REQUIRE(why == nullptr, "ldmk_igemm: tile_cfg=%d splitk=%d (warp-specialised bf16x3 tile) cannot run this problem: %s", cfg, wsk,
                 why ? why : "");
    LDMK_REQUIRE(!a.raw_slabs || wsk >= 2, "ldmk_igemm: raw_slabs needs a split-K plan");
    if (wsk > 1) {
      const long long b = a.batch > 1 ? a.batch : 1;
      LDMK_REQUIRE_MEM(a.splitk_ws && b * wsk * (long long)a.M * a.N <= a.splitk_ws_elems,
                       "ldmk_igemm: splitk=%d needs a workspace of %lld floats (ldmk_igemm_workspace_elems), %lld given", wsk,
                       b * wsk * (long long)a.M * a.N, a.splitk_ws ? a.splitk_ws_elems : 0LL);
    }
    if (!launch) return LDMK_OK;
    return igemm_ws_dispatch(a, wcfg, wsk, a.splitk_ws, (hipStream_t)stream);
  }
  if (cfg > kNumCfg + kNumRCfg) {      // slab GEMM: wave-autonomous, K split over the waves of a workgroup and over workgroups
    const int scfg = cfg - kNumCfg - kNumRCfg - 1;
    int ssk = a.splitk > 0 ? a.splitk : 1;
    const char* why = sgemm_unsupported(a, scfg, ssk);
    LDMK_REQUIRE(why == nullptr, "ldmk_igemm: tile_cfg=%d splitk=%d (slab GEMM) cannot run this problem: %s", cfg, ssk, why ? why : "");
    if (ssk > 1)
      LDMK_REQUIRE_MEM(a.splitk_ws && ssk * (long long)a.M * a.N <= a.splitk_ws_elems,
                       "ldmk_igemm: splitk=%d needs a workspace of %lld floats (ldmk_igemm_workspace_elems), %lld given", ssk,
                       ssk * (long long)a.M * a.N, a.splitk_ws ? a.splitk_ws_elems : 0LL);
    if (!launch) return LDMK_OK;
    return sgemm_dispatch(a, scfg, ssk, a.splitk_ws, (hipStream_t)stream);
  }
  if (cfg > kNumCfg) {      // row GEMM: one wave per output tile, K never split
    const char* why = rgemm_unsupported(a, cfg - kNumCfg - 1);
    LDMK_REQUIRE(why == nullptr, "ldmk_igemm: tile_cfg=%d (row GEMM) cannot run this problem: %s", cfg, why ? why : "");
    if (!launch) return LDMK_OK;
    return rgemm_dispatch(a, cfg - kNumCfg - 1, (hipStream_t)stream);
  }
  if (a.splitk > 0) sk = a.splitk;
  if (a.epi == LDMK_EPI_GEGLU && !a.raw_slabs) sk = 1;
  LDMK_REQUIRE(!a.raw_slabs || (sk >= 2 && a.batch <= 1 && !a.splitk_counters && !a.stats_out),
               "ldmk_igemm: raw_slabs needs a split-K plan (splitk >= 2), no batching / in-launch combine / stats_out");
  if (sk > 1 && a.splitk_counters) {
    const long long tiles = (long long)(a.batch > 1 ? a.batch : 1) * ((a.M + 63) / 64) * ((a.N + 63) / 64);   // smallest tile: 64x64
    LDMK_REQUIRE_MEM(tiles <= a.splitk_counters_len, "ldmk_igemm: in-launch split-K combine needs %lld zeroed counters, %d given",
                     tiles, a.splitk_counters_len);
    LDMK_REQUIRE((long long)sk * a.M * a.N * 4 < (1LL << 31), "ldmk_igemm: in-launch split-K combine addresses the %d slabs of one "
                 "batch entry with 32-bit offsets (%lld bytes)", sk, (long long)sk * a.M * a.N * 4);
  }
  if (sk > 1) {
    const long long b = a.batch > 1 ? a.batch : 1;
    LDMK_REQUIRE_MEM(a.splitk_ws && b * sk * (long long)a.M * a.N <= a.splitk_ws_elems,
                     "ldmk_igemm: splitk=%d needs a workspace of %lld floats (ldmk_igemm_workspace_elems), %lld given", sk,
                     b * sk * (long long)a.M * a.N, a.splitk_ws ? a.splitk_ws_elems : 0LL);
  }
  if (!launch) return LDMK_OK;
  hipStream_t st = (hipStream_t)stream;
  return a.b_trans ? dispatch<true>(a, cfg, sk, a.splitk_ws, st) : dispatch<false>(a, cfg, sk, a.splitk_ws, st);
}
