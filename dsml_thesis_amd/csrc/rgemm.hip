// Row GEMM for token-row Linear / 1x1 layers with small weights: wave-autonomous, no LDS, no barrier.
//
//   out[M][N] = epilogue( transform(A)[M][K] * W[K][N] )        (same contract as igemm.hip, rows mode)
//
// Why a second kernel family.  The f32 matrix pipe is slow relative to everything around it: one
// v_mfma_f32_32x32x2_f32 keeps a SIMD busy for 64 cycles and needs two operand registers.  A Linear on token
// rows with K = 160..640 is over after 5..20 staged K-slices per workgroup, so the LDS-tiled igemm kernel spends
// a third to a half of such a launch in its prologue (first global loads), its two barriers per slice and its
// epilogue, all of them in lockstep across the workgroup (profiles/r01_layers64_v4.txt: 41-78 TFLOP/s at K = 160).
// Here every wave is on its own:
//   * a wave owns a (32 TM) x (32 TN) output tile and walks K alone; nothing is shared, nothing is waited for
//     except its own loads, so the waves of a SIMD drift apart and one wave's prologue / epilogue runs under the
//     others' MFMAs (the SIMD arbitrates oldest-first: the older wave finishes early, its stores overlap);
//   * the A fragment comes straight from global memory: lane (m = lane & 31, h = lane >> 5) loads the float4
//     A[row0 + m][8 kb + 4 h .. + 3] and feeds its four components to the four MFMAs of that 8-deep K block.
//     The MFMA's k index is only a summation order, so "lane half h holds k = 8 kb + 4 h + s at step s" is as
//     good as the canonical "k = 2 s + h", provided B uses the same assignment;
//   * B is pre-packed once per weight into that fragment order (ldmk_pack_wfrag):
//         Wf[kb][nb][h][n][s] = W[8 kb + 4 h + s][32 nb + n]
//     so one wave-wide float4 load (1 KiB, contiguous) is the B operand of four MFMAs.  The weights of these
//     layers are 0.1-1.6 MB: they live in L2, and all waves of a launch walk the same column block at about the
//     same time;
//   * LayerNorm / GroupNorm-affine are applied to the A registers; bias, per-sample vector, residual, GEGLU and
//     the GroupNorm partial records of the output are the epilogue, as in igemm.hip.
// Results are deterministic (fixed k order per tile shape); the K-summation order differs from the LDS-tiled
// kernel's only in the order of the 8 addends inside each 8-deep block.
#include "ldmk_common.h"
#include <type_traits>

// Diagnostic build only (tools/rgemm_probe.hip defines LDMK_RG_STAMPS): per-wave s_memtime stamps around the
// prologue / main loop / epilogue go to args.splitk_ws (unused by this kernel), [wave][4] 64-bit ticks.
#ifdef LDMK_RG_STAMPS
#define RG_STAMP(i) do { if (lane == 0) reinterpret_cast<unsigned long long*>(p.splitk_ws)[(long long)wid * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RG_STAMP(i) do { } while (0)
#endif

namespace ldmk {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// global_load_dwordx4 with an immediate byte offset (13-bit signed).  hipcc does not count this load: every use of `d`
// sits behind wait_all() below.
template <int OFF>
__device__ __forceinline__ void gload4(f32x4& d, const void* ptr) {
  asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(ptr), "i"(OFF) : "memory");
}

// s_waitcnt vmcnt(0) that names every register the outstanding loads write ("+v"): no consumer can be scheduled above it
template <int TM, int TN>
__device__ __forceinline__ void wait_all(f32x4 (&a)[TM], f32x4 (&b)[TN], f32x4 (&t)[2]) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(a[i]));
#pragma unroll
  for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(b[j]));
  asm volatile("" : "+v"(t[0]));
  asm volatile("" : "+v"(t[1]));
}

// hipcc does not know that the asm statements below are MFMAs, so it pads no hazard in front of them.  The A operands
// are written by the compiler's own VALU code (LayerNorm / affine prologue), which it is free to sink between the MFMA
// statements right in front of the one that reads a component.  This statement takes every A register read-write: all of
// the prologue arithmetic is scheduled above it, and its s_nop covers the VALU-write -> MFMA-read wait states once.
template <int TM>
__device__ __forceinline__ void operands_ready(f32x4 (&a)[TM]) {
  if constexpr (TM == 1)
    asm volatile("s_nop 1" : "+v"(a[0]));
  else
    asm volatile("s_nop 1" : "+v"(a[0]), "+v"(a[1]));
}

__device__ __forceinline__ void mfma_asm(f32x16& c, float av, float bv) {
  asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c) : "v"(av), "v"(bv));
}

template <int TM, int TN, int TF>
__global__ __launch_bounds__(256) void rgemm_kernel(const ldmk_igemm_args p, const float4* __restrict__ wf) {
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const int tiles_m = (p.M + 32 * TM - 1) / (32 * TM);
  const int NB = p.N >> 5;
  const int tiles_n = NB / TN;
  const int wid = xcd_remap(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6);
  if (wid >= tiles_m * tiles_n) return;
  RG_STAMP(0);
  // (Measured and dropped: s_setprio by hardware wave slot, so that one wave per SIMD runs ahead and its store burst
  // overlaps the others' loops -- K=160 launches 3 % faster, the N >= 480 ones 2-20 % slower.)
  const int tn = wid / tiles_m, tm = wid - tn * tiles_m;      // m fastest: neighbours share the B column block
  const int row0 = tm * 32 * TM, nb0 = tn * TN;
  const int KB = p.K >> 3;
  const int kb_split = p.c0 >> 3;                              // first 8-deep block that reads the second source

  const float* arow0[TM];
  const float* arow1[TM];
  float mu[TM], rs[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int r = min(row0 + 32 * i + l31, p.M - 1);
    arow0[i] = p.a0 + (long long)r * p.c0 + 4 * half;
    arow1[i] = p.a1 ? p.a1 + (long long)r * p.c1 + 4 * half : nullptr;
    if (TF == LDMK_TF_LAYERNORM) {
      mu[i] = p.row_stats[2 * (long long)r];
      rs[i] = p.row_stats[2 * (long long)r + 1];
    }
  }
  const int sample = row0 / p.rows_per_sample;                 // a tile never straddles samples (checked by the host)
  const float* coef = TF == LDMK_TF_AFFINE ? p.tf_coef + (long long)sample * 2 * p.K + 4 * half : nullptr;
  const f32x4* bp = reinterpret_cast<const f32x4*>(wf) + (long long)nb0 * 64 + lane;   // + (kb * NB + j) * 64

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- main loop: hand-placed instruction stream (inline asm).
  // Two operand register sets; while block kb's 4 TM TN MFMAs run, the loads of block kb+1 (A fragments, gamma/beta or
  // scale/shift, TN B fragments -- in the order they will be consumed) are issued ONE PER MFMA behind its first
  // MFMAs, and the rest of the MFMAs follow with nothing else to issue.  Why by hand: a non-MFMA vector instruction
  // costs the SIMD an issue slot, and hipcc bunches the loads and their address arithmetic in front of the MFMA group
  // (or sinks them in front of their first use, whatever sched_group_barrier asks for once the loop has control flow).
  // Measured with s_memtime stamps on the compiler-scheduled form: 80 cycles per MFMA with one wave per SIMD, 213
  // with two, 303 with three -- co-resident waves serialised their bunches instead of overlapping them.  Front-loading
  // leaves 4 TM TN - loads MFMAs (~800 cycles at 1x5) between the last load and the `s_waitcnt vmcnt(0)` that opens
  // the next block: an L2 hit's worth.  Loads are unconditional (the last block re-loads itself).
  // hipcc neither counts these loads nor pads hazards around the asm MFMAs: the waits are explicit, operands_ready()
  // fences the VALU prologue, and the accumulators are read only after the drain behind the loop.
  constexpr int NT = (TF == LDMK_TF_NONE || TF == LDMK_TF_LAYERNORM_FOLDED) ? 0 : 2;
  constexpr int NLD = TM + NT + TN;                              // loads per block
  static_assert(4 * TM * TN >= NLD, "one load per MFMA");
  f32x4 a[2][TM], b[2][TN], t[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) t[q][0] = t[q][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* tp0 = nullptr;                                    // per-k prologue operands of block kb at tp + 8 kb
  const float* tp1 = nullptr;
  if (TF == LDMK_TF_LAYERNORM) { tp0 = p.ln_gamma + 4 * half; tp1 = p.ln_beta + 4 * half; }
  if (TF == LDMK_TF_AFFINE) { tp0 = coef; tp1 = coef + p.K; }

  const int nsrc = p.a1 ? 2 : 1;     // the channel concat of a skip connection = two passes over the same accumulators
  RG_STAMP(1);
  for (int src = 0; src < nsrc; ++src) {
    const int k0 = src ? kb_split : 0, k1 = src ? KB : (p.a1 ? kb_split : KB);      // this source's blocks [k0, k1)
    const float* rowp[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) rowp[i] = src ? arow1[i] : arow0[i];

    // load slot q of block kb into register set S (q: 0..TM-1 A rows, then the NT prologue vectors, then TN B tiles)
    auto load_slot = [&](auto S, auto Q, int kb) {
      constexpr int set = decltype(S)::value, q = decltype(Q)::value;
      if constexpr (q < TM) {
        gload4<0>(a[set][q], rowp[q] + 8 * (kb - k0));
      } else if constexpr (q < TM + NT) {
        gload4<0>(t[set][q - TM], (q - TM ? tp1 : tp0) + 8 * kb);
      } else {
        constexpr int j = q - TM - NT;
        gload4<j * 1024 - 2048>(b[set][j], reinterpret_cast<const char*>(bp + (long long)kb * NB * 64) + 2048);
      }
    };
    // one block: wait for set S, prologue VALU, MFMAs on set S with the loads of block `kn` into set 1-S behind them
    auto block = [&](auto S, int kn) {
      constexpr int set = decltype(S)::value;
      wait_all(a[set], b[set], t[set]);
      if (TF == LDMK_TF_LAYERNORM) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) a[set][i][c] = (a[set][i][c] - mu[i]) * rs[i] * t[set][0][c] + t[set][1][c];
      } else if (TF == LDMK_TF_AFFINE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int c = 0; c < 4; ++c) a[set][i][c] = fmaf(a[set][i][c], t[set][0][c], t[set][1][c]);
      }
      operands_ready(a[set]);
      static_for<0, 4 * TM * TN>([&](auto Q) {
        constexpr int q = decltype(Q)::value;
        constexpr int s = q / (TM * TN), j = (q / TM) % TN, i = q % TM;   // k-step major, then column tile, then row tile
        mfma_asm(acc[i][j], a[set][i][s], b[set][j][s]);
        if constexpr (q < NLD) load_slot(std::integral_constant<int, 1 - set>{}, Q, kn);
      });
    };
    static_for<0, NLD>([&](auto Q) { load_slot(std::integral_constant<int, 0>{}, Q, k0); });
    for (int kb = k0; kb < k1; kb += 2) {
      block(std::integral_constant<int, 0>{}, min(kb + 1, k1 - 1));
      if (kb + 1 < k1) {
        block(std::integral_constant<int, 1>{}, min(kb + 2, k1 - 1));
      } else {          // odd block count: the loads issued into set 1 are never used, but they must land before set 1 is reused
        wait_all(a[1], b[1], t[1]);
      }
    }
    if ((k1 - k0) % 2 == 0) wait_all(a[0], b[0], t[0]);          // the tail re-load of the last block (never consumed)
  }
  // the last MFMAs are still in the pipe (16 passes): pad before the compiler's epilogue code reads the accumulators
  // (the accumulators are operands of the statement: nothing that reads them may be scheduled above it)
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(acc[i][j]));

  RG_STAMP(2);
  // ---- epilogue.  C/D map: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 half.  Element offsets are 32-bit
  // (the host checks M * ldc < 2^31): one add per address instead of 64-bit pointer arithmetic per register.
  const float alpha = p.alpha;
  const int col0 = nb0 * 32;
  const int rlane = row0 + 4 * half;
  // LayerNorm folded through the product (ldmk.h): the loop above ran on raw rows
  constexpr bool lnf = TF == LDMK_TF_LAYERNORM_FOLDED;
  const float2* __restrict__ stats2 = reinterpret_cast<const float2*>(p.row_stats);
  float* __restrict__ outp = p.out;
  // Vector-memory operations complete in order on this ISA (one vmcnt for loads and stores): a tile that loads its
  // residual, waits, and stores also waits for the PREVIOUS tile's stores -- a load round trip plus a store round trip per
  // tile, 4-10 tiles per wave (tools/rgemm_probe.hip: epilogue 19-43k cycles against a 33-43k cycle main loop at K = 160).
  // So: per-column constants and the folded-LayerNorm row statistics are loaded once up front, and the residual of tile
  // t+1 is requested before tile t is stored (two register sets), which leaves only younger stores behind each wait.
  float2 st[lnf ? TM : 1][16];
  if constexpr (lnf) {                                        // (mean, rstd) of the 16 rows per row tile this lane holds
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) st[i][r] = stats2[min(rlane + i * 32 + (r & 3) + 8 * (r >> 2), p.M - 1)];
  }
  if (p.epi == LDMK_EPI_GEGLU) {
    if constexpr (TN % 2 == 0) {
      float bv[TN / 2], bg[TN / 2], csv[TN / 2], csg[TN / 2];
#pragma unroll
      for (int j = 0; j < TN; j += 2) {
        const int cv = col0 + j * 32 + l31, cg = cv + 32;       // packed (value | gate) 32-column pair
        bv[j / 2] = p.bias ? p.bias[cv] : 0.f;
        bg[j / 2] = p.bias ? p.bias[cg] : 0.f;
        csv[j / 2] = lnf ? p.ln_colsum[cv] : 0.f;
        csg[j / 2] = lnf ? p.ln_colsum[cg] : 0.f;
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
          const unsigned obase = (unsigned)rlane * p.ldc + ((col0 + j * 32) >> 1) + l31;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
            if (rlane + dr < p.M) {
              float v = acc[i][j][r] * alpha, g = acc[i][j + 1][r] * alpha;
              if constexpr (lnf) {
                v = fmaf(-st[i][r].x, csv[j / 2], v) * st[i][r].y;
                g = fmaf(-st[i][r].x, csg[j / 2], g) * st[i][r].y;
              }
              v += bv[j / 2];
              g += bg[j / 2];
              const float ge = gelu_erf_f(g);                                            // exact (erf) GELU
              outp[obase + (unsigned)(dr * p.ldc)] = v * ge;
            }
          }
        }
    }
    RG_STAMP(3);
    return;
  }
  const float* bvec = p.batch_vec ? p.batch_vec + (long long)sample * p.batch_vec_ld : nullptr;
  const float* __restrict__ resp = p.residual;
  float bias[TN], vec[TN], cs[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = col0 + j * 32 + l31;
    bias[j] = p.bias ? p.bias[col] : 0.f;
    vec[j] = bvec ? bvec[col] : 0.f;
    cs[j] = lnf ? p.ln_colsum[col] : 0.f;
  }
  // (the folded-LayerNorm instantiation keeps one set: its registers hold the row statistics, and none of the layers that
  // use it has a residual)
  constexpr int RSETS = lnf ? 1 : 2;
  float rv[RSETS][16];                                          // residual of the tile in work / of the next one
  // FULL (wave-uniform: the wave's 32 TM rows lie inside M) and HASR / HASV (a residual / a per-sample vector is present) are
  // compile-time in the body below: no row predicate around every store, no operand branches between them (round 5, as in the
  // lean epilogues of igemm.hip / igemm_ps.hip; same arithmetic, same order)
  auto finish = [&](auto FULL_, auto HASR_, auto HASV_) {
    constexpr bool FULL = decltype(FULL_)::value, HASR = decltype(HASR_)::value, HASV = decltype(HASV_)::value;
    auto load_residual = [&](auto T) {
      constexpr int t = decltype(T)::value, i = t / TN, j = t % TN;
      const unsigned obase = (unsigned)rlane * p.ldc + col0 + j * 32 + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr0 = i * 32 + (r & 3) + 8 * (r >> 2);
        const int dr = FULL ? dr0 : min(dr0, p.M - 1 - rlane);
        rv[t % RSETS][r] = resp[obase + (unsigned)(dr * p.ldc)];
      }
    };
    if constexpr (RSETS == 2 && HASR) load_residual(std::integral_constant<int, 0>{});
    static_for<0, TM * TN>([&](auto T) {
      constexpr int t = decltype(T)::value, i = t / TN, j = t % TN;
      const int col = col0 + j * 32 + l31;
      const unsigned obase = (unsigned)rlane * p.ldc + col;
      if constexpr (HASR) {
        if constexpr (RSETS == 1) load_residual(T);
        else if constexpr (t + 1 < TM * TN) load_residual(std::integral_constant<int, t + 1>{});
      }
      float vals[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int dr = i * 32 + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] * alpha;
        if constexpr (lnf) v = fmaf(-st[i][r].x, cs[j], v) * st[i][r].y;
        v += bias[j];                                             // same association as igemm.hip
        if constexpr (HASV) v += vec[j];
        if constexpr (HASR) v += rv[t % RSETS][r];
        vals[r] = v;
        if (FULL || rlane + dr < p.M) outp[obase + (unsigned)(dr * p.ldc)] = v;
      }
      if (p.stats_out && (FULL || row0 + i * 32 < p.M)) {
        const float shift = __shfl(vals[0], l31, 64);           // row 0 of the 32-row tile
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
    });
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  const bool full = row0 + 32 * TM <= p.M;
  if (full && resp && !bvec) finish(T_{}, T_{}, F_{});
  else if (full && !resp && !bvec) finish(T_{}, F_{}, F_{});
  else if (resp && bvec) finish(F_{}, T_{}, T_{});
  else if (resp) finish(F_{}, T_{}, F_{});
  else if (bvec) finish(F_{}, F_{}, T_{});
  else finish(F_{}, F_{}, F_{});
  RG_STAMP(3);
}

// W[K][ldb] (row-major, N used columns) -> Wf[K/8][N/32][2][32][4] (see the header comment)
__global__ __launch_bounds__(256) void pack_wfrag_kernel(const float* __restrict__ w, int ldb, int K, int N,
                                                         float4* __restrict__ wf) {
  const int NB = N >> 5;
  const long long total = (long long)(K >> 3) * NB * 64;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long long t = i >> 6;
    const int nb = (int)(t % NB), kb = (int)(t / NB);
    const int n = nb * 32 + (lane & 31), k = kb * 8 + 4 * (lane >> 5);
    float4 v;
    v.x = w[(long long)k * ldb + n]; v.y = w[(long long)(k + 1) * ldb + n];
    v.z = w[(long long)(k + 2) * ldb + n]; v.w = w[(long long)(k + 3) * ldb + n];
    wf[i] = v;
  }
}

// operands of LDMK_TF_LAYERNORM_FOLDED (ldmk.h): thread <-> column, rows walked in order (coalesced across the wave)
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const float* __restrict__ w, int ldb, int K, int N,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ bias, float* __restrict__ w_out,
                                                             float* __restrict__ colsum, float* __restrict__ bias_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double cs = 0.0, bb = bias ? (double)bias[n] : 0.0;
  for (int k = 0; k < K; ++k) {
    const float x = w[(long long)k * ldb + n];
    const float y = gamma[k] * x;
    w_out[(long long)k * N + n] = y;
    cs += (double)y;                 // the sum of what the matrix cores will multiply, not of the unrounded products
    bb += (double)beta[k] * (double)x;
  }
  colsum[n] = (float)cs;
  bias_out[n] = (float)bb;
}

struct RTile { int tm, tn; };
static const RTile kRTiles[] = {{1, 5}, {2, 5}, {1, 4}, {2, 4}, {1, 2}, {1, 1}};
constexpr int kNumRTiles = sizeof(kRTiles) / sizeof(kRTiles[0]);

template <int TM, int TN>
static int launch_r(const ldmk_igemm_args& a, hipStream_t st) {
  const int tiles = ((a.M + 32 * TM - 1) / (32 * TM)) * ((a.N / 32) / TN);
  const dim3 grid((tiles + 3) / 4), block(256);
  const float4* wf = reinterpret_cast<const float4*>(a.w_frag);
  switch (a.a_tf) {
    case LDMK_TF_NONE: hipLaunchKernelGGL((rgemm_kernel<TM, TN, LDMK_TF_NONE>), grid, block, 0, st, a, wf); break;
    case LDMK_TF_LAYERNORM_FOLDED:        // raw rows in the loop, the two per-row scalars in the epilogue
      hipLaunchKernelGGL((rgemm_kernel<TM, TN, LDMK_TF_LAYERNORM_FOLDED>), grid, block, 0, st, a, wf);
      break;
    case LDMK_TF_AFFINE: hipLaunchKernelGGL((rgemm_kernel<TM, TN, LDMK_TF_AFFINE>), grid, block, 0, st, a, wf); break;
    default: hipLaunchKernelGGL((rgemm_kernel<TM, TN, LDMK_TF_LAYERNORM>), grid, block, 0, st, a, wf); break;
  }
  return check_launch("ldmk_igemm(rows)");
}

// Can tile configuration `rcfg` (0-based index into kRTiles) run this problem?  Returns a reason or nullptr.
const char* rgemm_unsupported(const ldmk_igemm_args& a, int rcfg) {
  if (rcfg < 0 || rcfg >= kNumRTiles) return "no such row-GEMM tile";
  const RTile t = kRTiles[rcfg];
  if (!a.w_frag) return "w_frag (ldmk_pack_wfrag) missing";
  if (a.a_mode != LDMK_A_ROWS || a.b_trans || a.batch > 1) return "rows mode, packed [K][N] weights, no batching";
  if (a.a_tf == LDMK_TF_AFFINE_SILU) return "GroupNorm+SiLU prologue is not built for the row GEMM";
  if (a.N % (32 * t.tn) != 0) return "N must be a multiple of the tile's columns";
  if (a.c0 % 8 != 0 || a.c1 % 8 != 0) return "channel counts must be multiples of 8";
  if ((long long)a.M * a.ldc >= (1LL << 31)) return "output exceeds 2^31 elements (32-bit epilogue offsets)";
  if (a.epi == LDMK_EPI_GEGLU && t.tn % 2 != 0) return "GEGLU needs (value, gate) tile pairs";
  if ((a.a_tf == LDMK_TF_AFFINE || a.batch_vec) && a.rows_per_sample % (32 * t.tm) != 0)
    return "per-sample operands need rows_per_sample to be a multiple of the tile's rows";
  return nullptr;
}

int rgemm_dispatch(const ldmk_igemm_args& a, int rcfg, hipStream_t st) {
  switch (rcfg) {
    case 0: return launch_r<1, 5>(a, st);
    case 1: return launch_r<2, 5>(a, st);
    case 2: return launch_r<1, 4>(a, st);
    case 3: return launch_r<2, 4>(a, st);
    case 4: return launch_r<1, 2>(a, st);
    default: return launch_r<1, 1>(a, st);
  }
}

}  // namespace ldmk

extern "C" int ldmk_fold_layernorm(const float* w, int ldb, int K, int N, const float* gamma, const float* beta,
                                   const float* bias, float* w_out, float* colsum, float* bias_out, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(w && gamma && beta && w_out && colsum && bias_out && K > 0 && N > 0 && ldb >= N, "ldmk_fold_layernorm: bad args");
  hipLaunchKernelGGL(fold_layernorm_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, ldb, K, N, gamma, beta,
                     bias, w_out, colsum, bias_out);
  return check_launch("ldmk_fold_layernorm");
}

extern "C" long long ldmk_wfrag_elems(int K, int N) { return (K % 8 == 0 && N % 32 == 0) ? (long long)K * N : -1; }

extern "C" int ldmk_pack_wfrag(const float* w, int ldb, int K, int N, float* wfrag, void* stream) {
  LDMK_ENTER();
  using namespace ldmk;
  LDMK_REQUIRE(w && wfrag && K > 0 && N > 0, "ldmk_pack_wfrag: bad args");
  LDMK_REQUIRE(K % 8 == 0 && N % 32 == 0 && ldb >= N, "ldmk_pack_wfrag: K%%8, N%%32, ldb>=N (K=%d N=%d ldb=%d)", K, N, ldb);
  const long long total = (long long)(K / 8) * (N / 32) * 64;
  long long g = (total + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(pack_wfrag_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream, w, ldb, K, N,
                     reinterpret_cast<float4*>(wfrag));
  return check_launch("ldmk_pack_wfrag");
}
