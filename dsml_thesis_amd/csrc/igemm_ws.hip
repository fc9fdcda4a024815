// Warp-specialised implicit GEMM for the fp32-accurate bf16x3 arithmetic (LDMK_COMPUTE_BF16X3, include/ldmk.h): tile_cfg 21 / 22.
//
// csrc/igemm.hip's bf16x3 form spends 60 % of a slice's time outside its matrix instructions (s_memtime stamps):
// every wave stages (global loads, three-way split, LDS stores), waits at two barriers and only then multiplies, and the
// 4x larger LDS / vector-memory traffic per useful FLOP of this arithmetic makes the staging as long as the products.  Two
// co-resident workgroups overlap only by accident.  Here the overlap is by construction:
//   * a workgroup is 8 waves, two per SIMD: waves 0-3 are CONSUMERS (each owns 64 rows x all BN columns of a 256 x BN tile
//     and does nothing but ds_read_b128 + v_mfma_f32_32x32x16_bf16), waves 4-7 are PRODUCERS (global loads of the next
//     16-deep K stage, GroupNorm / SiLU / LayerNorm prologue, the exact split of A into three bf16 images, LDS stores; the
//     pre-split weights are copied);
//   * two LDS stages of 16 k; ONE workgroup barrier per stage: producers fill stage s+1 while consumers multiply stage s;
//   * 256 rows per workgroup: the B tile (three images, the larger half of the staged bytes) is shared by twice the matrix
//     work of the 128-row tile.
// Same operands, K order (32-channel chunk major, tap minor), split-K slabs + reduce launch and epilogue (bias, per-sample
// vector, residual, folded LayerNorm, GEGLU, GroupNorm records) as igemm_kernel<BF = 3>, and every accumulator sees the same
// sequence of matrix instructions as in its 32-deep-slice tiles (tile_cfg 1 / 5): bitwise equal results at equal split-K.
#include "ldmk_common.h"

// Diagnostic build only (tools/ws_probe.hip defines LDMK_WS_STAMPS): per-wave cycle totals of the pipeline's phases go to
// args.splitk_counters as [workgroup][wave][4] 64-bit ticks (producers: store, load issue, barrier wait; consumers: products,
// -, barrier wait; [3] = total).
#ifdef LDMK_WS_STAMPS
#define WS_T(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); ws_acc[i] += t_ - ws_last; ws_last = t_; } while (0)
#else
#define WS_T(i) do { } while (0)
#endif

namespace ldmk {

typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ wbf16x4 ws_bf4(const float4& v) { return wbf16x4{(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w}; }
__device__ __forceinline__ void ws_split3(const float4& v, wbf16x4& h, wbf16x4& m, wbf16x4& l) {
  h = ws_bf4(v);
  const float4 r = make_float4(v.x - (float)h[0], v.y - (float)h[1], v.z - (float)h[2], v.w - (float)h[3]);
  m = ws_bf4(r);
  l = ws_bf4(make_float4(r.x - (float)m[0], r.y - (float)m[1], r.z - (float)m[2], r.w - (float)m[3]));
}

__device__ __forceinline__ wu32x4 ws_rsrc(const void* ptr, unsigned bytes) {      // raw buffer descriptor, uniform -> SGPRs
  const unsigned long long a = (unsigned long long)ptr;
  wu32x4 r;
  r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
  r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xFFFFu);
  r.z = __builtin_amdgcn_readfirstlane(bytes);
  r.w = 0x00020000u;
  return r;
}

constexpr int WS_BM = 256;
constexpr int WS_RS = 24;            // bf16 per LDS row: 16 k + 8 pad (48 B: 16 consecutive rows cover the 64 banks exactly once)
constexpr int WS_AR = 4;             // float4 of A per producer thread and stage (256 rows x 16 k / 256 producer threads)

template <int TN>
__global__ __launch_bounds__(512) void igemm_ws_kernel(const ldmk_igemm_args p, const int splitk, float* __restrict__ ws) {
  constexpr int TM = 2;
  constexpr int BN = 32 * TN;
  constexpr int AIMG = WS_BM * WS_RS, BIMG = BN * WS_RS;       // bf16 elements per image
  constexpr int STAGE = 3 * (AIMG + BIMG);
  constexpr int BITEMS = 3 * BN * 2;                             // 16-byte items of the three B images per stage
  constexpr int BSI = (BITEMS + 255) / 256;
  constexpr int NL = WS_AR + BSI;                                // loads per producer thread and stage
  extern __shared__ __attribute__((aligned(16))) __bf16 smem_ws[];   // [2][STAGE]

  // (the wave index as a SCALAR: the producer / consumer branches below contain workgroup barriers, so they must be real
  //  scalar branches, never exec-masked regions that a wave could walk through with an empty mask)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const bool producer = wave >= 4;
  const int ptid = tid & 255;
  const int wm = wave & 3;

  const int tiles_m = (p.M + WS_BM - 1) / WS_BM;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int bid = xcd_remap(blockIdx.x, tiles_m * tiles_n);
  const int m0 = (bid % tiles_m) * WS_BM;
  const int n0 = (bid / tiles_m) * BN;
  const int ks = blockIdx.y, bz = blockIdx.z;

  const float* __restrict__ a0 = p.a0 + (long long)bz * p.a_bstride;
  const float* __restrict__ a1 = p.a1;
  const int Cin = p.c0 + p.c1;
  const int nkc = p.K / 32;
  const int it_per = (nkc + splitk - 1) / splitk;
  const int it_begin = ks * it_per;
  const int it_end = min(nkc, it_begin + it_per);
  const int n16 = it_end > it_begin ? 2 * (it_end - it_begin) : 0;
  const bool conv = p.a_mode == LDMK_A_CONV3X3;
  const int tf = p.a_tf == LDMK_TF_LAYERNORM_FOLDED ? LDMK_TF_NONE : p.a_tf;

  f32x16 acc[TM][TN];
  auto ws_compute = [&](int st) {             // consumers: one 16-deep step, 6 TM TN matrix instructions
    const __bf16* Aw = smem_ws + st * STAGE + (wm * 64 + l31) * WS_RS + 8 * half;
    const __bf16* Bw = smem_ws + st * STAGE + 3 * AIMG + l31 * WS_RS + 8 * half;
    wbf16x8 a8[3][TM];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
      for (int i = 0; i < TM; ++i) a8[g][i] = *reinterpret_cast<const wbf16x8*>(Aw + g * AIMG + i * 32 * WS_RS);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      wbf16x8 b8[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) b8[g] = *reinterpret_cast<const wbf16x8*>(Bw + g * BIMG + j * 32 * WS_RS);
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        // smallest partial products first (images: 0 = hi, 1 = mid, 2 = lo) -- the order of igemm_kernel<BF = 3>
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[2][i], b8[0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][i], b8[1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[1][i], b8[0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[0][i], b8[0], acc[i][j], 0, 0, 0);
      }
    }
  };

  // ---- the pipeline.  Every wave executes exactly 1 + n16 workgroup barriers; the two roles run separate loops so that the
  // accumulators (consumers) and the staging registers (producers) share the register file instead of adding up.
#ifdef LDMK_WS_STAMPS
  unsigned long long ws_acc[3] = {0, 0, 0}, ws_last = __builtin_amdgcn_s_memtime();
  const unsigned long long ws_t0 = ws_last;
  unsigned long long* ws_dbg = reinterpret_cast<unsigned long long*>(p.splitk_counters) + ((long long)blockIdx.x * 8 + wave) * 4;
#endif
  if (producer) {                                 // pair g stages the local stages t = g, g + 2, ... (buffer t & 1 = g)
    // ---- producer bookkeeping: rows arow + 64 i, 4 consecutive k of the 16-deep stage
    const int arow = ptid >> 2, acol = (ptid & 3) * 4;        // rows arow + 64 i
    int r_n[WS_AR];
    unsigned r_mask[WS_AR], aoff0[WS_AR], aoff1[WS_AR];
    float ln_mean[WS_AR], ln_rstd[WS_AR];
  #pragma unroll
    for (int i = 0; i < WS_AR; ++i) {
      const int m = m0 + arow + 64 * i;
      const bool valid = m < p.M;
      const int mm = valid ? m : 0;
      r_n[i] = mm / p.rows_per_sample;
      long long pix = mm;
      unsigned mask = valid ? 1u : 0u;
      if (conv) {
        const int px = mm - r_n[i] * p.rows_per_sample;
        const int oy = px / p.out_w, ox = px - oy * p.out_w;
        const int y0 = oy * p.stride - p.pad_lo, x0 = ox * p.stride - p.pad_lo;
        mask = 0;
  #pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = y0 + t / 3, ix = x0 + t % 3;
          if (valid && iy >= 0 && ix >= 0 && iy < p.in_h && ix < p.in_w) mask |= 1u << t;
        }
        pix = ((long long)r_n[i] * p.in_h + y0) * p.in_w + x0;       // tap (0,0): outside the image for border rows (masked taps only)
      }
      r_mask[i] = mask;
      aoff0[i] = (unsigned)((pix * p.c0 + acol) * 4);                 // mod 2^32: exact for valid taps once the slice scalar is added
      aoff1[i] = (unsigned)((pix * p.c1 + acol) * 4);
      ln_mean[i] = ln_rstd[i] = 0.f;
      if (tf == LDMK_TF_LAYERNORM) {
        ln_mean[i] = p.row_stats[2 * (long long)mm];
        ln_rstd[i] = p.row_stats[2 * (long long)mm + 1];
      }
    }
    const long long samples_ = ((long long)p.M + p.rows_per_sample - 1) / p.rows_per_sample;
    const long long a_rows_ = conv ? samples_ * p.in_h * p.in_w : (long long)p.M;
    // buffer descriptors as four SGPRs each (the loads below are inline asm: the compiler does not count them, the waits are
    // placed by hand -- see the pipeline)
    const wu32x4 rs_a0 = ws_rsrc(a0, (unsigned)(a_rows_ * p.c0 * 4));
    const wu32x4 rs_a1 = ws_rsrc(a1 ? a1 : a0, (unsigned)(a_rows_ * p.c1 * 4));
    const __bf16* wx = reinterpret_cast<const __bf16*>(p.w_split) + (long long)bz * p.w_split_bstride;
    const wu32x4 rs_wx = ws_rsrc(wx, (unsigned)(3LL * p.N * p.w_split_ld * 2));
    unsigned bxoff[BSI], bxlds[BSI];
  #pragma unroll
    for (int i = 0; i < BSI; ++i) {
      const int idx = ptid + 256 * i;
      const int img = idx / (BN * 2), rem = idx - img * (BN * 2);
      const int nn = rem >> 1, q = rem & 1;
      const bool ok = idx < BITEMS && n0 + nn < p.N;
      bxoff[i] = ok ? (unsigned)((((long long)img * p.N + n0 + nn) * p.w_split_ld + q * 8) * 2) : 0xFFFFFFFFu;
      bxlds[i] = (unsigned)((3 * AIMG + img * BIMG + nn * WS_RS + q * 8) * 2);       // byte offset inside a stage
    }

    // Two register sets, stage t in set t & 1.  The loads of stage t are issued two barriers before the stage is split and
    // stored: a global load under this kernel's own traffic takes longer than one stage's products.  They are inline asm so
    // that the wait in front of a set's use can be s_waitcnt vmcnt(NL) -- "all but the NL youngest", i.e. everything except
    // the other set, issued one barrier later; vector-memory loads return in order.  (With compiler-counted loads the wait
    // it placed was vmcnt(0), which also waited for the younger set: 2650 cycles per stage in split+store, consumers idle
    // 1320 of 3500 -- s_memtime stamps, tools/ws_probe.hip.)  Every stage issues exactly NL loads (stages past the end read
    // out of range: zeros, no memory traffic), so the count is a constant.
    wu32x4 areg0[WS_AR], areg1[WS_AR], breg0[BSI], breg1[BSI];
    auto ws_load = [&](int s, wu32x4 (&areg)[WS_AR], wu32x4 (&breg)[BSI]) {     // issue the NL loads of local stage s
      const bool live = s < n16;
      const int kc = it_begin + (s >> 1), h16 = s & 1;
      int tap = 0, cc = kc;
      if (conv) { cc = kc / 9; tap = kc - cc * 9; }
      const bool second = cc * 32 >= p.c0;
      const int cs = second ? p.c1 : p.c0;
      const int dy = tap / 3, dx = tap - dy * 3;
      const unsigned sa = (unsigned)(((conv ? (dy * p.in_w + dx) * cs : 0) + cc * 32 - (second ? p.c0 : 0) + h16 * 16) * 4);
      const unsigned sb = (unsigned)((kc * 32 + h16 * 16) * 2);
      const unsigned tbit = 1u << tap;
      unsigned oa[WS_AR], ob[BSI];
  #pragma unroll
      for (int i = 0; i < WS_AR; ++i) oa[i] = (live && (r_mask[i] & tbit)) ? (second ? aoff1[i] : aoff0[i]) + sa : 0xFFFFFFFFu;
  #pragma unroll
      for (int i = 0; i < BSI; ++i) ob[i] = (live && bxoff[i] != 0xFFFFFFFFu) ? bxoff[i] + sb : 0xFFFFFFFFu;
      const wu32x4 rs_a = second ? rs_a1 : rs_a0;
  #pragma unroll
      for (int i = 0; i < WS_AR; ++i) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(areg[i]) : "v"(oa[i]), "s"(rs_a) : "memory");
  #pragma unroll
      for (int i = 0; i < BSI; ++i) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(breg[i]) : "v"(ob[i]), "s"(rs_wx) : "memory");
    };
    auto ws_wait = [&](wu32x4 (&areg)[WS_AR], wu32x4 (&breg)[BSI]) {            // the older set has landed (names its registers)
      static_assert(NL == 7 || NL == 8, "the wait below is written for 7 or 8 loads per stage");
      if constexpr (NL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  #pragma unroll
      for (int i = 0; i < WS_AR; ++i) asm volatile("" : "+v"(areg[i]));
  #pragma unroll
      for (int i = 0; i < BSI; ++i) asm volatile("" : "+v"(breg[i]));
    };
    auto ws_store = [&](int s, int st, wu32x4 (&araw)[WS_AR], wu32x4 (&breg)[BSI]) {   // prologue + split + LDS stores of stage s into buffer st
      __bf16* base = smem_ws + st * STAGE;
      float4 areg[WS_AR];
  #pragma unroll
      for (int i = 0; i < WS_AR; ++i)
        areg[i] = make_float4(__uint_as_float(araw[i].x), __uint_as_float(araw[i].y), __uint_as_float(araw[i].z), __uint_as_float(araw[i].w));
      if (tf != LDMK_TF_NONE) {
        const int kc = it_begin + (s >> 1), h16 = s & 1;
        int tap = 0, cc = kc;
        if (conv) { cc = kc / 9; tap = kc - cc * 9; }
        const int c = cc * 32 + h16 * 16 + acol;
        if (tf == LDMK_TF_LAYERNORM) {
          const float4 g4 = *reinterpret_cast<const float4*>(p.ln_gamma + c);
          const float4 b4 = *reinterpret_cast<const float4*>(p.ln_beta + c);
  #pragma unroll
          for (int i = 0; i < WS_AR; ++i) {
            float4 v = areg[i];
            const float mu = ln_mean[i], rs = ln_rstd[i];
            v.x = (v.x - mu) * rs * g4.x + b4.x; v.y = (v.y - mu) * rs * g4.y + b4.y;
            v.z = (v.z - mu) * rs * g4.z + b4.z; v.w = (v.w - mu) * rs * g4.w + b4.w;
            areg[i] = (r_mask[i] & 1u) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
          }
        } else {
  #pragma unroll
          for (int i = 0; i < WS_AR; ++i) {
            if ((r_mask[i] >> tap) & 1u) {     // padded taps stay exactly zero
              const float* cf = p.tf_coef + ((long long)r_n[i] * 2) * Cin + c;
              const float4 sc = *reinterpret_cast<const float4*>(cf);
              const float4 sh = *reinterpret_cast<const float4*>(cf + Cin);
              float4 v = areg[i];
              v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
              v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
              if (tf == LDMK_TF_AFFINE_SILU) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
              areg[i] = v;
            }
          }
        }
      }
  #pragma unroll
      for (int i = 0; i < WS_AR; ++i) {
        wbf16x4 h, m, l;
        ws_split3(areg[i], h, m, l);
        __bf16* d = base + (arow + 64 * i) * WS_RS + acol;
        *reinterpret_cast<wbf16x4*>(d) = h;
        *reinterpret_cast<wbf16x4*>(d + AIMG) = m;
        *reinterpret_cast<wbf16x4*>(d + 2 * AIMG) = l;
      }
  #pragma unroll
      for (int i = 0; i < BSI; ++i) {
        if (ptid + 256 * i < BITEMS) *reinterpret_cast<wu32x4*>(reinterpret_cast<char*>(base) + bxlds[i]) = breg[i];
      }
    };

    // Measured (s_memtime stamps, tools/ws_probe.hip; conv 160->160 at 64x64, B = 16): per 16-deep stage the consumers' 60 matrix
    // instructions take 2180 cycles (1920 when the pipe is theirs), the producers' ~125 vector / LDS / load instructions 2550 --
    // about 20 cycles each -- and the stage 3500: on one SIMD the two streams mostly take turns (an MFMA in flight holds the
    // vector register ports; the same finding as rgemm.hip's 80 / 213 / 303 cycles per MFMA at 1 / 2 / 3 waves per SIMD).
    // s_setprio(3) on the producers: no change (2577).  So specialisation buys the overlap of the MEMORY latency (no wave
    // ever waits for a load here) but not of the issue slots: this tile runs level with the two-workgroup LDS-tiled form
    // (179 vs 165 us on that convolution, 91 vs 92 us at M = 16384 N = 320 K = 1280, 89 vs 91 at M = 4096 N = 640 K = 2560)
    // and is chosen per shape by the autotuner like any other tile.  What would move both forms is fewer non-matrix
    // instructions per product: activations stored pre-split by their producers and copied global -> LDS without passing
    // through registers.
    // n16 is even (two stages per 32-deep chunk)
    ws_load(0, areg0, breg0);
    ws_load(1, areg1, breg1);
    ws_wait(areg0, breg0);
    if (n16 > 0) ws_store(0, 0, areg0, breg0);
    ws_load(2, areg0, breg0);
    __syncthreads();
    WS_T(2);
    for (int s = 0; s < n16; s += 2) {
      ws_wait(areg1, breg1);                       // stage s + 1 (< n16 always); the younger loads in flight are stage s + 2's
      WS_T(1);
      ws_store(s + 1, 1, areg1, breg1);
      WS_T(0);
      ws_load(s + 3, areg1, breg1);
      WS_T(1);
      __syncthreads();
      WS_T(2);
      ws_wait(areg0, breg0);
      WS_T(1);
      if (s + 2 < n16) ws_store(s + 2, 0, areg0, breg0);
      WS_T(0);
      ws_load(s + 4, areg0, breg0);
      WS_T(1);
      __syncthreads();
      WS_T(2);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the trailing out-of-range loads)
#ifdef LDMK_WS_STAMPS
    if (lane == 0) { ws_dbg[0] = ws_acc[0]; ws_dbg[1] = ws_acc[1]; ws_dbg[2] = ws_acc[2]; ws_dbg[3] = __builtin_amdgcn_s_memtime() - ws_t0; }
#endif
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  __syncthreads();
  WS_T(2);
  for (int s = 0; s < n16; ++s) {
    ws_compute(s & 1);
    WS_T(0);
    __syncthreads();
    WS_T(2);
  }
#ifdef LDMK_WS_STAMPS
  if (lane == 0) { ws_dbg[0] = ws_acc[0]; ws_dbg[1] = ws_acc[1]; ws_dbg[2] = ws_acc[2]; ws_dbg[3] = __builtin_amdgcn_s_memtime() - ws_t0; }
#endif

  // ---- epilogue (consumers).  C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const int rowbase = m0 + wm * 64;
  const int colbase = n0;
  if (splitk > 1) {   // raw partial slab [ks][M][N]; igemm_reduce_kernel (or the consumer, raw_slabs) sums them
    float* slab = ws + ((long long)bz * splitk + ks) * p.M * p.N;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = colbase + j * 32 + l31;
      if (col >= p.N) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          if (row < p.M) slab[(long long)row * p.N + col] = acc[i][j][r];
        }
    }
    return;
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
            for (int r = 0; r < 16; ++r) st[r] = stats2[min(rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half, p.M - 1)];
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rowbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < p.M) {
              float v = acc[i][j][r] * alpha, g = acc[i][j + 1][r] * alpha;
              if (lnf) {      // same arithmetic as igemm.hip / rgemm.hip / igemm_reduce_kernel
                v = fmaf(-st[r].x, csv, v) * st[r].y;
                g = fmaf(-st[r].x, csg, g) * st[r].y;
              }
              v += bv;
              g += bg;
              outp[(long long)row * p.ldc + oc] = v * gelu_erf_f(g);
            }
          }
        }
      }
    }
    return;
  }
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
        // GroupNorm partial record of this 32-row tile x column (the record of gn_partial_kernel / igemm_kernel)
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

int launch_splitk_reduce(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st);

template <int TN>
static int ws_launch(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  constexpr int BN = 32 * TN;
  constexpr size_t lds = (size_t)2 * 3 * (WS_BM + BN) * WS_RS * 2;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(igemm_ws_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  const int tiles = ((a.M + WS_BM - 1) / WS_BM) * ((a.N + BN - 1) / BN);
  hipLaunchKernelGGL(igemm_ws_kernel<TN>, dim3(tiles, splitk, a.batch > 1 ? a.batch : 1), dim3(512), lds, st, a, splitk, ws);
  if (splitk > 1 && !a.raw_slabs) return launch_splitk_reduce(a, splitk, ws, st);
  return check_launch("ldmk_igemm(ws)");
}

// wcfg 0: 256 x 160 (this UNet's channel counts), 1: 256 x 128 (GEGLU value/gate pairs)
const char* igemm_ws_unsupported(const ldmk_igemm_args& a, int wcfg, int splitk) {
  if (a.compute != LDMK_COMPUTE_BF16X3 || !a.w_split) return "the warp-specialised tiles run the bf16x3 arithmetic only (compute, w_split)";
  if (a.b_trans) return "b_trans";
  if (a.upsample) return "upsampling folded into the gather";
#ifndef LDMK_WS_STAMPS
  if (a.splitk_counters) return "in-launch split-K combine";
#endif
  if (a.skip_a0) return "fused skip connection";
  if (a.epi == LDMK_EPI_GEGLU && (wcfg != 1 || splitk > 1)) return "GEGLU needs the 256x128 tile and no split-K";
  if (splitk > a.K / 32) return "more K slices than 32-deep chunks";
  return nullptr;
}
int igemm_ws_dispatch(const ldmk_igemm_args& a, int wcfg, int splitk, float* ws, hipStream_t st) {
  return wcfg == 1 ? ws_launch<4>(a, splitk, ws, st) : ws_launch<5>(a, splitk, ws, st);
}

}  // namespace ldmk
