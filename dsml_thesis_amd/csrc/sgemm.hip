// Slab GEMM: the implicit GEMM of igemm.hip for SMALL row counts (batch 1-2: M = 64 .. 2048 rows), built for latency.
//
//   out[M][N] = epilogue( transform(A)[M][K] * W[K][N] )        (same contract as igemm.hip; tile_cfg 13..20)
//
// Why a third kernel family.  At batch 1 (the reference's shipped talking-face mode, progressive_sampling_difftalk.py:282-317)
// the step's 145 GEMMs are 73 % of its time (tools/marginal_cost.py: 2.58 of 3.51 ms, 17.8 us per GEMM incl. its reduce
// launch) although they execute 39 GFLOP -- 0.25 ms of matrix work.  The LDS-tiled kernel walks K in 32-deep slices, each a
// dependent chain  global load -> LDS store -> barrier -> MFMA  shared by four waves: with M = 64 rows there are few
// tiles, so K is split 6-64 ways and every workgroup still pays that chain several times, behind a launch ramp.
// Here every wave is on its own again (like rgemm.hip), and K is split twice:
//   * a wave owns a (32 TM) x (32 TN) output tile for ONE K range and keeps the whole range's operands in flight: A
//     fragments straight from global memory (rows mode: float4 per lane; 3x3 convolution: the im2col gather as a clamped
//     float4 load + select, tap validity from a 9-bit per-row mask), B from the fragment-order weight copy
//     (ldmk_pack_wfrag: one contiguous 1-KiB wave load per four MFMAs), two register sets of 4 eight-deep blocks;
//   * the NW (4, 8 or 16) waves of a workgroup take consecutive K ranges of the same tile and are summed through LDS in a
//     fixed binary tree (NW = 4: (w0 + w2) + (w1 + w3)), each lane reading back exactly the slots it would have
//     written: no layout change.  16 waves = 4 per SIMD: while one waits for its operands the others multiply, which is
//     what hides the HBM latency of a weight matrix that is read exactly once;
//   * `splitk` workgroups per tile split K further; their partial tiles go to slabs [splitk][M][N] that the reduce
//     launch (igemm.hip) or the consumer (ldmk_post, raw_slabs) sums in slab order.  With splitk = 1 wave 0 runs the
//     epilogue itself (bias, per-sample vector, residual, folded LayerNorm, GEGLU, GroupNorm records).
// A 64 x 640 x 5760 convolution (8x8 level) becomes 320 workgroups x 4 waves with 11 eight-deep blocks each: every SIMD of
// the chip streams its own 1/1280 of the 14.7 MB weight matrix, once.
// Deterministic: the k order inside a wave, the tree and the slab order are all fixed.
#include "ldmk_common.h"
#include <type_traits>

// Diagnostic build only (tools/sgemm_probe.py builds a second library with -DLDMK_SG_STAMPS): per-wave s_memrealtime stamps
// (constant 100 MHz) at the phase boundaries go to args.stats_out as [workgroup][wave][8] 64-bit ticks.
#ifdef LDMK_SG_STAMPS
#define SG_STAMP(i) do { if (lane == 0) reinterpret_cast<unsigned long long*>(p.stats_out)[((long long)blockIdx.x * NW + wave) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define SG_STAMP(i) do { } while (0)
#endif

namespace ldmk {

// eight-deep K blocks per register set (two sets): as many as the register file of the workgroup shape allows
// (the GroupNorm-affine prologue loads two more vectors per block: half the depth)
template <int TM, int TN, int NW, int TF>
constexpr int sg_u() { return (NW >= 16 ? 4 : (TM * TN == 1 ? 8 : (TM * TN == 2 ? 6 : 4))) / (TF == LDMK_TF_AFFINE ? 2 : 1); }

template <int TM, int TN, int U>
struct SgFrag {
  f32x4 a[U][TM];
  f32x4 b[U][TN];
  f32x4 t[U][2];                   // scale / shift of the GroupNorm-affine prologue
  unsigned ok[U];                  // convolution: bit i set <=> row tile i's tap lies inside the image (else the operand is 0)
};

template <int TM, int TN, int NW, bool CONV, int TF>
__global__ __launch_bounds__(64 * NW) void sgemm_kernel(const ldmk_igemm_args p, const float4* __restrict__ wf, const int splitk,
                                                        float* __restrict__ ws) {
  constexpr int SG_U = sg_u<TM, TN, NW, TF>();
  __shared__ float red[NW / 2][TM * TN * 16 * 64];
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  // (readfirstlane: the wave index -- hence the whole K-block arithmetic below -- lives in scalar registers)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  SG_STAMP(0);
  const int tiles_m = (p.M + 32 * TM - 1) / (32 * TM);
  const int NB = p.N >> 5;
  // XCD-aware order (workgroups are dealt round-robin over the 8 XCDs, each with its own L2): consecutive remapped ids
  // share an XCD, and they walk  K slab (slowest) > column tile > row tile, so the row tiles that read the SAME weight
  // block sit in one L2 (the block comes from HBM once, not once per row tile) and an XCD touches only its slabs' A columns
  const int tiles_all = tiles_m * ((p.N >> 5) / TN);
  const int rid = xcd_remap(blockIdx.x, gridDim.x);
  const int kslab = rid / tiles_all, tile = rid - kslab * tiles_all;
  const int tn = tile / tiles_m, tm = tile - tn * tiles_m;      // m fastest: neighbours share the B column block
  const int row0 = tm * 32 * TM, nb0 = tn * TN;
  const int KB = p.K >> 3;
  // this wave's K range: part q of NW * splitk equal parts (in eight-deep blocks)
  const int q = kslab * NW + wave, parts = NW * splitk;
  // (32-bit: KB * parts < 2^31 is checked by the host)
  const int kb0 = (int)((unsigned)(KB * q) / (unsigned)parts), kb1 = (int)((unsigned)(KB * (q + 1)) / (unsigned)parts);
  const f32x4* bp = reinterpret_cast<const f32x4*>(wf) + (long long)nb0 * 64 + lane;
  // The weights are what comes from HBM (each byte is read once per step): their loads of the first register set are issued
  // before anything else is computed, so that the A addressing below (pixel decode, tap masks) runs under their latency
  // (tools/sgemm_probe.py: 1.3 us of set-up in front of the first load, 2-2.5 us until the first operands arrive).
  SgFrag<TM, TN, SG_U> f0, f1;
  auto load_b = [&](SgFrag<TM, TN, SG_U>& f, int kb) {
#pragma unroll
    for (int u = 0; u < SG_U; ++u) {
      const int k = min(kb + u, kb1 - 1);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.b[u][j] = bp[((long long)k * NB + j) * 64];
    }
  };
  if (kb0 < kb1) load_b(f0, kb0);
  __builtin_amdgcn_sched_barrier(0);


  // ---- per-lane A addressing
  const float* abase0[TM];
  const float* abase1[TM];
  unsigned tapmask[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = min(row0 + 32 * i + l31, p.M - 1);
    if constexpr (CONV) {
      const int hw = p.out_h * p.out_w;                          // stride 1, pad 1: in == out
      const int n = r / hw, rem = r - n * hw;
      const int y = rem / p.out_w, x = rem - y * p.out_w;
      abase0[i] = p.a0 + ((long long)(n * p.in_h + y) * p.in_w + x) * p.c0 + 4 * half;
      abase1[i] = nullptr;
      unsigned m = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < p.in_h && xx >= 0 && xx < p.in_w) m |= 1u << t;
      }
      tapmask[i] = m;
    } else {
      abase0[i] = p.a0 + (long long)r * p.c0 + 4 * half;
      abase1[i] = p.a1 ? p.a1 + (long long)r * p.c1 + 4 * half : nullptr;
      tapmask[i] = 0;
    }
  }
  const int kb_split = p.c0 >> 3;                                // rows mode: first block of the second source
  // convolution with the ResBlock's 1x1 skip connection fused as extra K: blocks [kb_conv, KB) read rows of skip_a0 | skip_a1
  const int kb_conv = CONV ? (9 * p.c0) >> 3 : 0;
  const int kb_skip1 = kb_conv + (p.skip_c0 >> 3);
  const float* sbase0[TM];
  const float* sbase1[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = min(row0 + 32 * i + l31, p.M - 1);
    sbase0[i] = (CONV && p.skip_a0) ? p.skip_a0 + (long long)r * p.skip_c0 + 4 * half : nullptr;
    sbase1[i] = (CONV && p.skip_a1) ? p.skip_a1 + (long long)r * p.skip_c1 + 4 * half : nullptr;
  }
  const int sample = row0 / p.rows_per_sample;                   // a tile never straddles samples when it matters (host check)
  const float* coef = TF == LDMK_TF_AFFINE ? p.tf_coef + (long long)sample * 2 * p.K + 4 * half : nullptr;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A loads of the U blocks starting at kb (blocks past kb1 re-load the last one: never consumed)
  auto load_a = [&](SgFrag<TM, TN, SG_U>& f, int kb) {
#pragma unroll
    for (int u = 0; u < SG_U; ++u) {
      const int k = min(kb + u, kb1 - 1);
      if (CONV && k >= kb_conv) {                                  // (wave-uniform) fused skip connection: plain rows
        f.ok[u] = ~0u;
#pragma unroll
        for (int i = 0; i < TM; ++i)
          f.a[u][i] = *reinterpret_cast<const f32x4*>(k < kb_skip1 ? sbase0[i] + 8 * (k - kb_conv) : sbase1[i] + 8 * (k - kb_skip1));
      } else if constexpr (CONV) {
        // K order of ldmk_pack_conv3x3: [chunk of 32 channels][tap][32]  ->  k = (chunk * 9 + tap) * 4 + quarter
        const int ct = k >> 2, qq = k & 3;
        const int chunk = ct / 9, tap = ct - chunk * 9;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int off = (dy * p.in_w + dx) * p.c0 + chunk * 32 + qq * 8;
        // (the zero of a padded tap is selected in compute(), not here: a select on the loaded value would make the
        //  wave wait for the loads it has just issued, before the previous set's MFMAs -- no overlap at all)
        unsigned okm = 0;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const bool ok = (tapmask[i] >> tap) & 1u;
          f.a[u][i] = *reinterpret_cast<const f32x4*>(abase0[i] + (ok ? off : chunk * 32 + qq * 8));
          okm |= (ok ? 1u : 0u) << i;
        }
        f.ok[u] = okm;
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float* src = (k < kb_split || !abase1[i]) ? abase0[i] + 8 * k : abase1[i] + 8 * (k - kb_split);
          f.a[u][i] = *reinterpret_cast<const f32x4*>(src);
        }
      }
      if constexpr (TF == LDMK_TF_AFFINE) {
        f.t[u][0] = *reinterpret_cast<const f32x4*>(coef + 8 * k);
        f.t[u][1] = *reinterpret_cast<const f32x4*>(coef + p.K + 8 * k);
      }
    }
  };
  auto load = [&](SgFrag<TM, TN, SG_U>& f, int kb) { load_b(f, kb); load_a(f, kb); };
  auto compute = [&](SgFrag<TM, TN, SG_U>& f, int kb) {
#pragma unroll
    for (int u = 0; u < SG_U; ++u) {
      if (kb + u < kb1) {                                        // wave-uniform
        if constexpr (TF == LDMK_TF_AFFINE) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) f.a[u][i][c] = fmaf(f.a[u][i][c], f.t[u][0][c], f.t[u][1][c]);
        }
        if constexpr (CONV) {
#pragma unroll
          for (int i = 0; i < TM; ++i)
            if (!((f.ok[u] >> i) & 1u)) f.a[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[u][i][s], f.b[u][j][s], acc[i][j], 0, 0, 0);
      }
    }
  };
  SG_STAMP(1);
  if (kb0 < kb1) {
    load_a(f0, kb0);
    SG_STAMP(2);
    // (sched_barrier: nothing of a compute() -- its tap selects wait for the set's loads -- may be scheduled above the
    //  issue of the other set's loads, or the wave would sit out a memory round trip with no load in flight)
    for (int kb = kb0; kb < kb1; kb += 2 * SG_U) {
      if (kb + SG_U < kb1) load(f1, kb + SG_U);
      __builtin_amdgcn_sched_barrier(0);
      compute(f0, kb);
      if (kb + SG_U < kb1) {
        if (kb + 2 * SG_U < kb1) load(f0, kb + 2 * SG_U);
        __builtin_amdgcn_sched_barrier(0);
        compute(f1, kb + SG_U);
      }
    }
  }

  SG_STAMP(3);
  // ---- the four K ranges of the workgroup: ((w0 + w2) + (w1 + w3)), every lane re-reads its own slots
  auto put = [&](int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[slot][((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
  };
  auto add = [&](int slot) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += red[slot][((i * TN + j) * 16 + r) * 64 + lane];
  };
#pragma unroll
  for (int s = NW / 2; s >= 1; s >>= 1) {
    if (wave >= s && wave < 2 * s) put(wave - s);
    __syncthreads();
    if (wave < s) add(wave);
    if (s > 1) __syncthreads();                                  // the slots are rewritten in the next round
  }
  SG_STAMP(4);
  if (wave != 0) return;

  // ---- wave 0: slab store or epilogue.  C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half
  const int col0 = nb0 * 32;
  const int rlane = row0 + 4 * half;
  if (splitk > 1) {
    float* slab = ws + (long long)kslab * p.M * p.N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rlane + i * 32 + (r & 3) + 8 * (r >> 2);
          if (row < p.M) slab[(long long)row * p.N + col0 + j * 32 + l31] = acc[i][j][r];
        }
    SG_STAMP(5);
    return;
  }
  const float alpha = p.alpha;
  const bool lnf = p.a_tf == LDMK_TF_LAYERNORM_FOLDED;
  const float2* __restrict__ stats2 = reinterpret_cast<const float2*>(p.row_stats);
  if (p.epi == LDMK_EPI_GEGLU) {
    if constexpr (TN % 2 == 0) {
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const int cv = col0 + j * 32 + l31, cg = cv + 32;       // packed (value | gate) 32-column pair
        const float bv = p.bias ? p.bias[cv] : 0.f, bg = p.bias ? p.bias[cg] : 0.f;
        const float csv = lnf ? p.ln_colsum[cv] : 0.f, csg = lnf ? p.ln_colsum[cg] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = rlane + i * 32 + (r & 3) + 8 * (r >> 2);
            if (row < p.M) {
              float v = acc[i][j][r] * alpha, g = acc[i][j + 1][r] * alpha;
              if (lnf) {
                const float2 st = stats2[row];
                v = fmaf(-st.x, csv, v) * st.y;
                g = fmaf(-st.x, csg, g) * st.y;
              }
              v += bv;
              g += bg;
              p.out[(long long)row * p.ldc + ((col0 + j * 32) >> 1) + l31] = v * gelu_erf_f(g);
            }
          }
      }
    }
    return;
  }
  const float* bvec = p.batch_vec ? p.batch_vec + (long long)sample * p.batch_vec_ld : nullptr;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32 + l31;
    const float bias = p.bias ? p.bias[col] : 0.f, vec = bvec ? bvec[col] : 0.f, cs = lnf ? p.ln_colsum[col] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float vals[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = min(rlane + i * 32 + (r & 3) + 8 * (r >> 2), p.M - 1);
        float v = acc[i][j][r] * alpha;
        if (lnf) {
          const float2 st = stats2[row];
          v = fmaf(-st.x, cs, v) * st.y;
        }
        v += bias;                                                // same association as igemm.hip / rgemm.hip
        if (bvec) v += vec;
        if (p.residual) v += p.residual[(long long)row * p.ldc + col];
        vals[r] = v;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rlane + i * 32 + (r & 3) + 8 * (r >> 2);
        if (row < p.M) p.out[(long long)row * p.ldc + col] = vals[r];
      }
      if (p.stats_out && row0 + i * 32 < p.M) {                   // GroupNorm partial record of this 32-row tile and column
        const float shift = __shfl(vals[0], l31, 64);
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
          float* d = p.stats_out + ((long long)((row0 + i * 32) >> 5) * p.N + col) * 3;
          d[0] = shift; d[1] = sm; d[2] = sq;
        }
      }
    }
  }
}

struct STile { int tm, tn, nw; };
static const STile kSTiles[] = {{2, 1, 4}, {2, 2, 4}, {1, 1, 4}, {1, 2, 4}, {1, 1, 8}, {1, 1, 16}, {2, 1, 8}, {1, 2, 8}};
constexpr int kNumSTiles = sizeof(kSTiles) / sizeof(kSTiles[0]);

int launch_splitk_reduce(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st);      // igemm.hip

template <int TM, int TN, int NW>
static int launch_s(const ldmk_igemm_args& a, int splitk, float* ws, hipStream_t st) {
  const int tiles = ((a.M + 32 * TM - 1) / (32 * TM)) * ((a.N / 32) / TN);
  const dim3 grid(tiles * splitk), block(64 * NW);
  const float4* wf = reinterpret_cast<const float4*>(a.w_frag);
  const bool conv = a.a_mode == LDMK_A_CONV3X3;
  const bool aff = a.a_tf == LDMK_TF_AFFINE;
  if (conv)
    hipLaunchKernelGGL((sgemm_kernel<TM, TN, NW, true, LDMK_TF_NONE>), grid, block, 0, st, a, wf, splitk, ws);
  else if (aff)
    hipLaunchKernelGGL((sgemm_kernel<TM, TN, NW, false, LDMK_TF_AFFINE>), grid, block, 0, st, a, wf, splitk, ws);
  else
    hipLaunchKernelGGL((sgemm_kernel<TM, TN, NW, false, LDMK_TF_NONE>), grid, block, 0, st, a, wf, splitk, ws);
  if (splitk > 1 && !a.raw_slabs) return launch_splitk_reduce(a, splitk, ws, st);
  return check_launch("ldmk_igemm(slab)");
}

// Can slab-GEMM tile `scfg` (0-based index into kSTiles) with this K split run the problem?  Returns a reason or nullptr.
const char* sgemm_unsupported(const ldmk_igemm_args& a, int scfg, int splitk) {
  if (scfg < 0 || scfg >= kNumSTiles) return "no such slab-GEMM tile";
  const STile t = kSTiles[scfg];
  if (!a.w_frag) return "w_frag (ldmk_pack_wfrag) missing";
  if (a.b_trans || a.batch > 1 || a.compute != LDMK_COMPUTE_F32) return "packed [K][N] fp32 weights, no batching";
  if (a.a_tf == LDMK_TF_AFFINE_SILU || a.a_tf == LDMK_TF_LAYERNORM) return "prologue not built for the slab GEMM (none / affine / folded layernorm)";
  if (a.N % (32 * t.tn) != 0) return "N must be a multiple of the tile's columns";
  if ((long long)a.M * a.ldc >= (1LL << 31)) return "output exceeds 2^31 elements";
  if (a.a_mode == LDMK_A_CONV3X3) {
    if (a.stride != 1 || a.upsample || a.pad_lo != 1 || a.a1 || a.in_h != a.out_h || a.in_w != a.out_w || a.a_tf != LDMK_TF_NONE)
      return "3x3 convolutions: stride 1, pad 1, one source, no prologue";
    if (a.skip_a0 && (a.skip_c0 % 8 != 0 || a.skip_c1 % 8 != 0)) return "skip channels must be multiples of 8";
  } else if (a.c0 % 8 != 0 || a.c1 % 8 != 0) {
    return "channel counts must be multiples of 8";
  }
  if (a.epi == LDMK_EPI_GEGLU && (t.tn % 2 != 0 || (splitk > 1 && !a.raw_slabs))) return "GEGLU needs (value, gate) tile pairs and no K split across workgroups";
  if ((a.a_tf == LDMK_TF_AFFINE || a.batch_vec) && a.rows_per_sample % (32 * t.tm) != 0)
    return "per-sample operands need rows_per_sample to be a multiple of the tile's rows";
  if (splitk < 1 || (long long)t.nw * splitk > (a.K >> 3)) return "every wave needs at least one eight-deep K block (K / 8 >= waves x splitk)";
  if ((long long)(a.K >> 3) * t.nw * splitk >= (1LL << 31)) return "K x splits exceeds the 32-bit range arithmetic";
  if (splitk > 1 && a.splitk_counters) return "the in-launch combine is not built for the slab GEMM";
  if (a.raw_slabs && splitk < 2) return "raw_slabs needs splitk >= 2";
  return nullptr;
}

int sgemm_dispatch(const ldmk_igemm_args& a, int scfg, int splitk, float* ws, hipStream_t st) {
  switch (scfg) {
    case 0: return launch_s<2, 1, 4>(a, splitk, ws, st);
    case 1: return launch_s<2, 2, 4>(a, splitk, ws, st);
    case 2: return launch_s<1, 1, 4>(a, splitk, ws, st);
    case 3: return launch_s<1, 2, 4>(a, splitk, ws, st);
    case 4: return launch_s<1, 1, 8>(a, splitk, ws, st);
    case 5: return launch_s<1, 1, 16>(a, splitk, ws, st);
    case 6: return launch_s<2, 1, 8>(a, splitk, ws, st);
    default: return launch_s<1, 2, 8>(a, splitk, ws, st);
  }
}

}  // namespace ldmk
